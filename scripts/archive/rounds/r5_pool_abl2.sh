#!/bin/bash
# Round 5: pooled-unit kernel, fp64 — what the x gathers cost (timing-only diagnostic libraries; results wrong by construction)
for rep in 1 2; do
for v in "" _pabl3 _pabl4 _pabl6; do
  echo "== variant '${v}' (rep $rep)"
  TILESPMV_LIB_VARIANT=$v python scripts/archive/rounds/r5_quick_time.py fem3_68,fem3_86 f64 2>&1 | grep -v amdgpu.ids
done
done
