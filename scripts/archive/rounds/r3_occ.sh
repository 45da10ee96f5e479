#!/bin/bash
# round-3: what do fewer resident workgroups cost (LDS padding builds), and waves per SIMD / trip size of the workgroup entry mode
out=gpurun_out/$1; mkdir -p $out; shift
V='"" TILESPMV_WG_STRIPS=16 TILESPMV_WG_STRIPS=16,LIB=_pad10 TILESPMV_WG_STRIPS=16,LIB=_pad18 TILESPMV_WG_STRIPS=16,LIB=_w7 TILESPMV_WG_STRIPS=16,LIB=_w8 TILESPMV_WG_STRIPS=16,LIB=_ct4 TILESPMV_WG_STRIPS=16,LIB=_ct4w8 TILESPMV_WG_STRIPS=16,LIB=_ct5w7 TILESPMV_WG_STRIPS=16,LIB=_abl2 TILESPMV_WG_STRIPS=16,TILESPMV_STRIP_COST=1200 TILESPMV_WG_STRIPS=16,TILESPMV_STRIP_COST=2400'
for wl in ${@:-powerlaw8000000}; do
  eval timeout -k 10 500 python scripts/exp_bench.py $wl $V > $out/$wl.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl.txt | tail -13
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
