#!/bin/bash
# nontemporal streams as a plan rule (TILESPMV_NT_STREAM unset / 0 / 1) + diagnostic builds that also read descriptors (_ntd) and per-strip entries (_ntc) nontemporally
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ntv
for wl in ${@:-laplacian4096 lap3d256 nlpkkt160 laplacian2048 laplacian1448}; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "TILESPMV_NT_STREAM=0" "TILESPMV_NT_STREAM=1" "LIB=_ntd" "LIB=_ntc" "Q=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/rule_$wl.txt
done
echo "== nlpkkt160 f64"
EXP_F64=1 timeout -k 10 400 python scripts/exp_bench.py nlpkkt160 "Q=1" "TILESPMV_NT_STREAM=0" "TILESPMV_NT_STREAM=1" "LIB=_ntd" "LIB=_ntc" "Q=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/rule_nlpkkt160_f64.txt
for wl in powerlaw8000000 powerlaw2000000 webbase scircuit; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "TILESPMV_NT_STREAM=0" "TILESPMV_NT_STREAM=1" "Q=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/rule_$wl.txt
done
