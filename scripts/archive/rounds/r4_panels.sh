#!/bin/bash
# round 4: column panels of the entry lists — parity, then unpanelled / chosen by timing / fixed passes on the irregular class
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4panels; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_panels or slab_paced" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 400 python tests/gpu_fuzz.py 30 9000 > $out/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $out/fuzz.log
export TILESPMV_PLAN_VERBOSE=1
for wl in ${WLS:-uniform8_4000000 uniform8_8000000 rmat22x8 bandrand4x3_2000000 powerlaw8000000 circuit1000000 webbase}; do
  timeout -k 10 600 python scripts/exp_bench.py $wl TILESPMV_X_PANEL_KB=0 Q=auto TILESPMV_X_PANEL_MERGE=1 TILESPMV_X_PANEL_MERGE=2 TILESPMV_X_PANEL_MERGE=4 TILESPMV_X_PANEL_MERGE=8 \
      TILESPMV_X_PANEL_KB=1024,TILESPMV_X_PANEL_MERGE=1 TILESPMV_X_PANEL_MERGE=2,TILESPMV_COO_ORDERED=0 > $out/exp_$wl.txt 2>&1
  echo "== $wl rc=$?"; grep -v "amdgpu.ids\|brick order" $out/exp_$wl.txt | cut -c1-200
done
