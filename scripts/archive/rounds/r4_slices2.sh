#!/bin/bash
# round 4: column slices on XCDs — tests + fuzz, then a kernel trace of the sliced / panelled / plain uniform 4 M plans (who takes the time: the unit kernel or the slice launches)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4slices2; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "column_slices or column_panels or irregular_class or every_plan_kind or tile_row_shards" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python tests/gpu_fuzz.py 120 13000 > $out/fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -1 $out/fuzz.log
[ $rc -eq 0 ] || exit $rc
cd /tmp
for v in 1 0; do
  TILESPMV_X_SLICE_PASSES=$v rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_s$v -- python $GRAFT_REPO_ROOT/bench.py --workload uniform8_4000000 --steps 30 --warmup 5 --no-extras > $GRAFT_REPO_ROOT/$out/bench_s$v.json 2> $GRAFT_REPO_ROOT/$out/bench_s$v.err
  echo "slice passes $v rc=$?"
  f=$(find $GRAFT_REPO_ROOT/$out/trace_s$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200
done
