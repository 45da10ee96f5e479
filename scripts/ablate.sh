#!/bin/bash
# timing-only ablations of k_units (needs the temporary TILESPMV_ABLATE hooks; results are wrong by construction)
for m in 0 1 2 4 3 6 7; do
  echo -n "ablate=$m : "; TILESPMV_ABLATE=$m python scripts/exp_bench.py laplacian4096 "" 2>&1 | grep -v amdgpu.ids
done
