#!/bin/bash
# One parametrised driver for "the same workloads under a list of exp_bench.py variants" (replaces round 3's r3_ntv.sh … r3_ntv5.sh, which differed only in these lists):
#   scripts/exp_sweep.sh <out tag> "<workload> [<workload> ...]" <variant> [<variant> ...]
# a workload may carry a dtype suffix (nlpkkt160:f64); a variant is exp_bench.py's syntax (ENV=v,ENV=v,LIB=_build).  Results: gpurun_out/<tag>/<workload>.txt.
# The round-3 sweeps as calls of this script:
#   exp_sweep.sh r3ntv  "laplacian4096 lap3d256 nlpkkt160 nlpkkt160:f64" Q=1 LIB=_ntv TILESPMV_XCD_CHUNK=1024 TILESPMV_XCD_CHUNK=4096 TILESPMV_XCD_CHUNK=128 LIB=_ntv,TILESPMV_XCD_CHUNK=1024
#   exp_sweep.sh r3ntv  "powerlaw8000000 webbase scircuit band40_2000000 laplacian1024" Q=1 LIB=_ntv LIB=_nte Q=2 LIB=_ntv,Q=3 LIB=_nte,Q=4
#   exp_sweep.sh r3rule "laplacian4096 lap3d256 nlpkkt160 nlpkkt160:f64 laplacian2048 laplacian1448" Q=1 TILESPMV_NT_STREAM=0 TILESPMV_NT_STREAM=1 LIB=_ntd LIB=_ntc Q=2
#   exp_sweep.sh r3rule "powerlaw8000000 powerlaw2000000 webbase scircuit" Q=1 TILESPMV_NT_STREAM=0 TILESPMV_NT_STREAM=1 Q=2
cd $GRAFT_REPO_ROOT
tag=$1; wls=$2; shift 2
mkdir -p gpurun_out/$tag
for spec in $wls; do
  wl=${spec%%:*}; dt=${spec#*:}
  echo "== $spec"
  ( [ "$dt" = "f64" ] && [ "$wl" != "$spec" ] && export EXP_F64=1; [ "$dt" = "f32" ] && [ "$wl" != "$spec" ] && export EXP_F32=1
    timeout -k 10 600 python scripts/exp_bench.py $wl "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/$(echo $spec | tr ':' '_').txt )
done
