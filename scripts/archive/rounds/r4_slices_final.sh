#!/bin/bash
# round 4: after the column slices — whole GPU suite, fuzz, the driver-shaped bench line, traffic of the sliced uniform plan
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4slicesf; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/gpu_fuzz.py 200 15000 > $out/fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -1 $out/fuzz.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
bash scripts/profile_traffic.sh r4s_uniform4m uniform8_4000000 f64 > $out/traffic_uniform4m.log 2>&1; echo "traffic rc=$?"; tail -3 $out/traffic_uniform4m.log
