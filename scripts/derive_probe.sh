#!/bin/bash
# Round 6 probe: shifted units take the previous unit's x rotated by one lane instead of gathering (timing-only library _abl11)
for v in "" _abl11; do
  echo "== variant '$v'"
  TILESPMV_LIB_VARIANT=$v python3 scripts/knob_time.py laplacian4096 f64 "" 2>&1 | grep -v amdgpu.ids
  TILESPMV_LIB_VARIANT=$v python3 scripts/knob_time.py lap3d256 f64 "" 2>&1 | grep -v amdgpu.ids
done
