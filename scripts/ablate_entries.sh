#!/bin/bash
# Timing-only ablations of the workgroup entry phase (k_units<.., 2>) on entry-heavy matrices.  Needs the diagnostic builds:
#   for v in 1 2 3 5; do make -C tilespmv_amd/csrc libs VARIANT=_abl$v EXTRA_DEFS=-DTILESPMV_ABL=$v; done
#   make -C tilespmv_amd/csrc libs VARIANT=_ct4 EXTRA_DEFS=-DWCOO_HEAVY_CT=4   (and _ct8, _w4 = -DECOO2_MIN_WAVES=4)
# ok=False is expected for the _abl builds (results are wrong by construction).
out=gpurun_out/${1:-abl}; mkdir -p $out; shift
for wl in ${@:-powerlaw8000000}; do
  timeout -k 10 500 python scripts/exp_bench.py $wl "" LIB=_abl1 LIB=_abl2 LIB=_abl3 LIB=_abl5 LIB=_ct4 LIB=_ct8 LIB=_w4 TILESPMV_COO_ORDERED=0 TILESPMV_STRIP_COST=800 > $out/$wl.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl.txt | tail -12
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
