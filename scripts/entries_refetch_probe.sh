#!/bin/bash
# Round 6 probe: is config 4's x re-fetch caused by the per-strip entry lists (their gathers touch the strip's own lines of x long before / after its units do)?
bash scripts/fetch_ab.sh laplacian4096 f64 "- _abl8 _abl9 _abl10" 2>&1 | grep -E "FETCH|failed"
for v in "" _abl9 _abl10; do
  echo "== variant '$v'"
  TILESPMV_LIB_VARIANT=$v python3 scripts/knob_time.py laplacian4096 f64 "" 2>&1 | grep -v amdgpu.ids
done
