#!/bin/bash
# nontemporal value loads (diagnostic build _ntv: make VARIANT=_ntv EXTRA_DEFS=-DTILESPMV_NT_VALUES=1 libs) and one-slab-per-XCD maps on the stencil workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ntv
for wl in ${@:-laplacian4096 lap3d256 nlpkkt160}; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "LIB=_ntv" "TILESPMV_XCD_CHUNK=1024" "TILESPMV_XCD_CHUNK=4096" "TILESPMV_XCD_CHUNK=128" "LIB=_ntv,TILESPMV_XCD_CHUNK=1024" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/$wl.txt
done
echo "== nlpkkt160 f64"
EXP_F64=1 timeout -k 10 400 python scripts/exp_bench.py nlpkkt160 "Q=1" "LIB=_ntv" "TILESPMV_XCD_CHUNK=1024" "TILESPMV_XCD_CHUNK=4096" "TILESPMV_XCD_CHUNK=128" "LIB=_ntv,TILESPMV_XCD_CHUNK=1024" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/nlpkkt160_f64.txt
