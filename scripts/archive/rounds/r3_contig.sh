#!/bin/bash
# plan arena blocks from hipExtMallocWithFlags(hipDeviceMallocContiguous) (TILESPMV_ARENA_FLAGS=4) against plain hipMalloc: instance spread, three instances each, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3contig
V='Q=1 TILESPMV_ARENA_FLAGS=4,Q=1 Q=2 TILESPMV_ARENA_FLAGS=4,Q=2 Q=3 TILESPMV_ARENA_FLAGS=4,Q=3'
for wl in ${@:-nlpkkt160 laplacian4096 lap3d256 powerlaw8000000}; do
  echo "== $wl"; timeout -k 10 600 python scripts/exp_bench.py $wl $V 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3contig/$wl.txt
done
echo "== nlpkkt160 f64"; EXP_F64=1 timeout -k 10 600 python scripts/exp_bench.py nlpkkt160 $V 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3contig/nlpkkt160_f64.txt
