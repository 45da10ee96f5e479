#!/bin/bash
# multi-vector kernel at nvec 8 on config 4 under plan knobs (one process per variant)
cd $GRAFT_REPO_ROOT
for v in "Q=1" "TILESPMV_MV_XCD_CHUNK=8" "TILESPMV_MV_XCD_CHUNK=16" "TILESPMV_MV_XCD_CHUNK=64" "TILESPMV_MV_XCD_CHUNK=128" "TILESPMV_STRIP_COST=400" "TILESPMV_STRIP_COST=200" "TILESPMV_X_WINDOW=2" "TILESPMV_X_WINDOW=2 TILESPMV_BRICK_ROWS=8" "TILESPMV_NT_STREAM=0"; do
  echo "== $v"
  ( export $v; timeout -k 10 300 python scripts/spmm_bench.py ${1:-laplacian4096} f64 8 2>&1 | grep -v amdgpu.ids | cut -c90-300 )
done
