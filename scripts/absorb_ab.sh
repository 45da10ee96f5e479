#!/bin/bash
# Round 6: absorbed list entries on / off over the bench workloads (default plans, fp64 and fp32), one session, two repetitions
for rep in 1 2; do
for v in 0 1; do
  echo "== TILESPMV_ABSORB=$v (rep $rep)"
  TILESPMV_ABSORB=$v python3 scripts/quick_time.py laplacian4096,lap3d256,nlpkkt160,band40_2000000,bandrand4x3_2000000,powerlaw8000000,shell4_780,road3400,tri2200s4096,circuit4m,webbase,scircuit,fem3_68 both 2>&1 | grep -v amdgpu.ids
done
done
