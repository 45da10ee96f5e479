#!/bin/bash
# round 4: slab-paced entry phase — parity first (new GPU test + fuzz seeds that force pacing), then unpaced / calibrated / fixed timetables on the irregular class
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4pace; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_paced or knobs_through_options" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 400 python tests/gpu_fuzz.py 40 7000 > $out/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $out/fuzz.log
export TILESPMV_PLAN_VERBOSE=1
for wl in ${WLS:-bandrand4x3_2000000 uniform8_8000000 powerlaw8000000 rmat22x8}; do
  timeout -k 10 600 python scripts/exp_bench.py $wl TILESPMV_PACE=0 TILESPMV_PACE=1 TILESPMV_PACE=1,TILESPMV_PACE_SLAB_KB=512 TILESPMV_PACE=1,TILESPMV_PACE_SLAB_KB=2048 TILESPMV_PACE=1,TILESPMV_PACE_WINDOW=3 \
      TILESPMV_PACE=1,TILESPMV_PACE_WINDOW=1 TILESPMV_PACE=1,TILESPMV_PACE_TEAM=96 TILESPMV_PACE=1,TILESPMV_PACE_TEAM=384 TILESPMV_PACE=1,TILESPMV_COO_ORDERED=0 TILESPMV_PACE=0,TILESPMV_COO_ORDERED=0 > $out/exp_$wl.txt 2>&1
  echo "== $wl rc=$?"; grep -v "amdgpu.ids\|brick order" $out/exp_$wl.txt | cut -c1-200
done
