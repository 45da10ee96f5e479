#!/bin/bash
# nontemporal value loads (_ntv) and value + entry-record loads (_nte) on the irregular / cache-resident / dense workloads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ntv
for wl in ${@:-powerlaw8000000 webbase scircuit band40_2000000 laplacian1024}; do
  echo "== $wl"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "LIB=_ntv" "LIB=_nte" "Q=2" "LIB=_ntv,Q=3" "LIB=_nte,Q=4" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ntv/$wl.txt
done
