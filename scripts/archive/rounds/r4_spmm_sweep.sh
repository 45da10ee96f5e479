#!/bin/bash
# multi-vector kernel at nvec 8 on config 4 (and the KKT stand-in) under plan knobs, round-4 kernels (one process per variant)
cd $GRAFT_REPO_ROOT
for wl in laplacian4096 nlpkkt160; do
for v in "Q=1" "TILESPMV_MV_XCD_CHUNK=8" "TILESPMV_MV_XCD_CHUNK=64" "TILESPMV_MV_XCD_CHUNK=128" "TILESPMV_MV_XCD_CHUNK=256" "TILESPMV_STRIP_COST=400" "TILESPMV_X_WINDOW=2" "TILESPMV_X_WINDOW=2 TILESPMV_MV_XCD_CHUNK=64" "TILESPMV_NT_STREAM=0" "TILESPMV_PLACEMENT_TRIES=1"; do
  echo "== $wl $v"
  ( export $v; timeout -k 10 300 python scripts/spmm_bench.py $wl f64 8 2>&1 | grep -v amdgpu.ids | cut -c90-300 )
done; done
