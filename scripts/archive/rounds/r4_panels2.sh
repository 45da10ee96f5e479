#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r4panels2; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_panels" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for wl in uniform8_4000000 uniform8_8000000 rmat22x8 bandrand4x3_2000000; do
  timeout -k 10 600 python scripts/exp_bench.py $wl TILESPMV_X_PANEL_KB=0 TILESPMV_X_PANEL_MERGE=2 TILESPMV_X_PANEL_MERGE=4 TILESPMV_X_PANEL_MERGE=8 Q=auto > $out/exp_$wl.txt 2>&1
  echo "== $wl"; grep -v "amdgpu.ids\|brick order" $out/exp_$wl.txt | cut -c1-160
done
