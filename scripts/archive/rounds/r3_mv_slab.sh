#!/bin/bash
# entry slab of the multi-vector kernel (TILESPMV_MV_SLAB=0 off / unset = rule), one process per run
cd $GRAFT_REPO_ROOT
for spec in "nlpkkt160 f32 2,4,8" "nlpkkt160 f64 2,8" "laplacian4096 f64 4,8" "laplacian4096 f32 8" "lap3d255 f64 2,8" "laplacian4095 f64 8" "scircuit f64 2,4,8"; do
  for slab in 0 -1; do
    echo "== $spec slab=$slab"
    ( [ $slab = 0 ] && export TILESPMV_MV_SLAB=0; timeout -k 10 300 python scripts/spmm_bench.py $spec 2>&1 | grep -v amdgpu.ids | cut -c90-700 )
  done
done
