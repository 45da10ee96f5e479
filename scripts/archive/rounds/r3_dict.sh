#!/bin/bash
# 4-B dictionary descriptors against the 12-B form (TILESPMV_DESC_DICT=0), same process, interleaved rounds.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3dict
for wl in ${@:-laplacian4096 lap3d256 nlpkkt160}; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "TILESPMV_DESC_DICT=0" "Q=2" "TILESPMV_DESC_DICT=0,Q=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3dict/$wl.txt
done
echo "== nlpkkt160 f64"
EXP_F64=1 timeout -k 10 400 python scripts/exp_bench.py nlpkkt160 "Q=1" "TILESPMV_DESC_DICT=0" "Q=2" "TILESPMV_DESC_DICT=0,Q=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3dict/nlpkkt160_f64.txt
