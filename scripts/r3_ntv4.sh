#!/bin/bash
# several plan instances per variant, alternating creation order: separates the nontemporal-stream effect from the instance (placement) effect
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ntv
for wl in ${@:-laplacian4096 nlpkkt160}; do
  echo "== $wl"
  timeout -k 10 500 python scripts/exp_bench.py $wl "TILESPMV_NT_STREAM=0,Q=1" "TILESPMV_NT_STREAM=1,Q=1" "LIB=_ntd,Q=1" "TILESPMV_NT_STREAM=0,Q=2" "TILESPMV_NT_STREAM=1,Q=2" "LIB=_ntd,Q=2" "TILESPMV_NT_STREAM=0,Q=3" "TILESPMV_NT_STREAM=1,Q=3" "LIB=_ntd,TILESPMV_NT_STREAM=0,Q=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/inst_$wl.txt
done
