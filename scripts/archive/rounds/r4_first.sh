#!/bin/bash
# round 4, first GPU session: parity after the options-mirror fix, gather locality sweep, counters of the two matrices nobody had looked at
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r4first
python -m pytest tests -m gpu -x -q > gpurun_out/r4first/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4first/pytest_gpu.log
timeout -k 10 500 python scripts/archive/rounds/r4_gather_locality.py 4000000 8 > gpurun_out/r4first/locality.txt 2>&1; echo "locality rc=$?"; cat gpurun_out/r4first/locality.txt
for wl in bandrand4x3_2000000 uniform8_8000000; do
  timeout -k 10 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r4first/bench_$wl.json 2> gpurun_out/r4first/bench_$wl.err; echo "bench $wl rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/r4first/bench_$wl.json'));print(d['ms_per_step'], d['roofline']['frac'], d['config']['entry_mode'], d['config']['strip_cost'], d['config']['tasks'], d['roofline']['plan_stream_bytes_per_launch'])"
done
scripts/pmc_short.sh r4_bandrand2m --workload bandrand4x3_2000000
scripts/pmc_short.sh r4_uniform8m --workload uniform8_8000000
