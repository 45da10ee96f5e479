#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof summary.  Usage: scripts/gpu_check.sh <tag>
set -o pipefail
tag=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/pytest_gpu_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_$tag.log
python bench.py --steps 200 --warmup 50 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"; cat gpurun_out/bench_$tag.json
