#!/bin/bash
for m in 0 16 24 18 22 23; do
  echo -n "ablate=$m : "; TILESPMV_ABLATE=$m python scripts/exp_bench.py laplacian4096 TILESPMV_UNIT_BATCH=4 2>&1 | grep -v amdgpu.ids | sed 's/TILESPMV_UNIT_BATCH=4 *//'
done
