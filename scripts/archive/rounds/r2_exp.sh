#!/bin/bash
# round-2 experiment batch: entry modes (per strip / per wavefront / per workgroup, column-ordered) x strip cost
out=gpurun_out/$1; mkdir -p $out; shift
wls=${@:-scircuit webbase powerlaw8000000 laplacian4096 lap3d256 nlpkkt160}
for wl in $wls; do
  timeout -k 10 400 python scripts/exp_bench.py $wl TILESPMV_WAVE_COO=0 TILESPMV_WAVE_COO=1 TILESPMV_WAVE_COO=2 TILESPMV_WAVE_COO=2,TILESPMV_STRIP_COST=800 TILESPMV_WAVE_COO=2,TILESPMV_STRIP_COST=1600 TILESPMV_WAVE_COO=2,TILESPMV_STRIP_COST=200 TILESPMV_WAVE_COO=2,TILESPMV_COO_NT=1 "" > $out/$wl.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl.txt | tail -9
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
