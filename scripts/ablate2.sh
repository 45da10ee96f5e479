#!/bin/bash
for m in 7 6 0; do for x in 0 1 3; do
  echo -n "ablate=$m remap=$x : "; TILESPMV_ABLATE=$m python scripts/exp_bench.py laplacian4096 TILESPMV_XCD_REMAP=$x,TILESPMV_XCD_CHUNK=32 2>&1 | grep -v amdgpu.ids | sed 's/TILESPMV_XCD[A-Z_=0-9,]* *//'
done; done
