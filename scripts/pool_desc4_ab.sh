#!/bin/bash
# Round 6: 4-byte pooled-dictionary descriptors (default) against the 8-byte pairs (TILESPMV_DESC_DICT=2), one session, two repetitions
for rep in 1 2; do
for v in "" 2; do
  echo "== TILESPMV_DESC_DICT='${v}' (rep $rep)"
  TILESPMV_DESC_DICT=$v python3 scripts/quick_time.py fem3_68,fem6_46,fem3_86,tet150,shell4_780 both 2>&1 | grep -v amdgpu.ids
done
done
