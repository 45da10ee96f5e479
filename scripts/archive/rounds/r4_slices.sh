#!/bin/bash
# round 4: column slices pinned to XCDs (k_entries_xcd) — parity, then the plain launch / the timed choice / fixed slice passes / the panelled choice on the irregular class
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4slices; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_slices or column_panels" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
export TILESPMV_PLAN_VERBOSE=1
for wl in ${WLS:-uniform8_4000000 uniform8_8000000 rmat22x8 bandrand4x3_2000000 powerlaw8000000}; do
  timeout -k 10 600 python scripts/exp_bench.py $wl TILESPMV_X_PANEL_KB=0 Q=auto TILESPMV_X_SLICE_PASSES=0 TILESPMV_X_SLICE_PASSES=1 TILESPMV_X_SLICE_PASSES=2 TILESPMV_X_SLICE_PASSES=4 \
      TILESPMV_X_PANEL_KB=1024,TILESPMV_X_SLICE_PASSES=2 TILESPMV_X_PANEL_KB=512,TILESPMV_X_SLICE_PASSES=1 > $out/exp_$wl.txt 2>&1
  echo "== $wl rc=$?"; grep -v "amdgpu.ids\|brick order" $out/exp_$wl.txt | cut -c1-200
done
