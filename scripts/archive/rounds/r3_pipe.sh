#!/bin/bash
# round-3: software-pipelined workgroup trips (next trip's records behind the current gathers) x trip size x waves per SIMD
out=gpurun_out/$1; mkdir -p $out; shift
V='TILESPMV_WG_STRIPS=16 TILESPMV_WG_STRIPS=16,LIB=_pipe0 TILESPMV_WG_STRIPS=16,LIB=_p1ct4 TILESPMV_WG_STRIPS=16,LIB=_p1ct4w7 TILESPMV_WG_STRIPS=16,LIB=_p1ct4w8 TILESPMV_WG_STRIPS=16,LIB=_p1ct3w8 TILESPMV_WG_STRIPS=16,LIB=_p1w5 TILESPMV_WG_STRIPS=16,LIB=_p1ct4,TILESPMV_STRIP_COST=2400 TILESPMV_WG_STRIPS=32,LIB=_p1ct4'
for wl in ${@:-powerlaw8000000}; do
  eval timeout -k 10 500 python scripts/exp_bench.py $wl $V > $out/$wl.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl.txt | tail -10
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
