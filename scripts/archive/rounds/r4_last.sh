#!/bin/bash
# round 4, last session: whole GPU suite, smoke, the default bench line as the driver runs it, then the evidence passes for every bench workload
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4last; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err ) 2> $out/bench_driver_args.time; echo "bench rc=$?"; tail -3 $out/bench_driver_args.time
( time timeout -k 10 900 python bench.py > $out/bench_default_args.json 2> $out/bench_default_args.err ) 2> $out/bench_default_args.time; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --data real --no-extras --no-cpu-baseline > $out/bench_real.json 2> $out/bench_real.err; echo "bench real rc=$?"
scripts/archive/rounds/r4_final.sh > $out/final.log 2>&1; grep "^{\|calls" $out/final.log | cut -c1-200
