#!/bin/bash
# where the nontemporal-stream rule should switch: mid-size workloads around the Infinity Cache size, rule off / on
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ntv
for wl in ${@:-laplacian2200 laplacian2400 laplacian2896 laplacian3400 powerlaw3000000 powerlaw4000000 powerlaw5000000 lap3d128 lap3d160}; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "TILESPMV_NT_STREAM=0" "TILESPMV_NT_STREAM=1" "TILESPMV_NT_STREAM=0,Q=2" "TILESPMV_NT_STREAM=1,Q=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/size_$wl.txt
done
