#!/bin/bash
# small irregular matrices: per-wavefront (1) against per-workgroup (2) entry lists around the rule's switch (est. workgroups < 768 -> mode 1)
cd $GRAFT_REPO_ROOT
for wl in powerlaw100000 powerlaw300000 powerlaw500000 powerlaw1000000 circuit100000 circuit300000 circuit600000 scircuit webbase; do
  echo "== $wl"; timeout -k 10 300 python scripts/exp_bench.py $wl "Q=1" "TILESPMV_WAVE_COO=1" "TILESPMV_WAVE_COO=2,TILESPMV_COO_ORDERED=0" "TILESPMV_WAVE_COO=2,TILESPMV_COO_ORDERED=1" 2>&1 | grep -v amdgpu.ids
done
