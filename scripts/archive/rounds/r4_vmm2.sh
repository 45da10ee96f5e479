#!/bin/bash
# round 4: chunked physical backing of large blocks as the default (64 MB) — the moved-plan test with forced chunking, then ten KKT plans (and six of band hbw 40) per setting in one process:
# hipMalloc blocks without / with the placement retry, 64-MB chunks without / with the retry
export TMPDIR=/tmp
out=gpurun_out/r4vmm2; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "survives_being_moved or capturable or column_slices" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for wl in "nlpkkt160 10" "band40_2000000 6" "nlpkkt160 f32 6"; do
  for mb in 0 64; do
    echo "== $wl  TILESPMV_ARENA_VMM_MB=$mb"
    TILESPMV_ARENA_VMM_MB=$mb timeout -k 10 500 python scripts/archive/rounds/r4_placement.py $wl 2>&1 | grep -v amdgpu.ids | cut -c1-220
  done
done
