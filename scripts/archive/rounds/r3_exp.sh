#!/bin/bash
# round-3 experiment batch: packed entry records, 16 / 32 strips per workgroup
out=gpurun_out/$1; mkdir -p $out; shift
for wl in ${@:-powerlaw8000000 webbase scircuit}; do
  timeout -k 10 500 python scripts/exp_bench.py $wl "" TILESPMV_WG_STRIPS=16 TILESPMV_WG_STRIPS=32 TILESPMV_WAVE_COO=2,TILESPMV_WG_STRIPS=32 TILESPMV_WAVE_COO=2,TILESPMV_WG_STRIPS=16 TILESPMV_WAVE_COO=1 TILESPMV_WAVE_COO=2,TILESPMV_WG_STRIPS=32,TILESPMV_COO_ORDERED=0 TILESPMV_WAVE_COO=2,TILESPMV_WG_STRIPS=32,TILESPMV_STRIP_COST=800 > $out/$wl.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl.txt | tail -9
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
