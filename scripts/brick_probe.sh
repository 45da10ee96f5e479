#!/bin/bash
# Round 6 probe: brick order with 8-row strips on config 4, brick shapes 2x8, 1x16, 16x1 (= linear), XCD window 32 — time and FETCH_SIZE
export TILESPMV_BRICK_ROWS=8 TILESPMV_PLAN_VERBOSE=1 TILESPMV_XCD_CHUNK=32
for sh in 1 3 4; do
  export TILESPMV_BRICK_SHAPE=$sh
  echo "== shape index $sh of {4x4, 2x8, 8x2, 1x16, 16x1}"
  python3 scripts/knob_time.py laplacian4096 f64 x_window=2,xcd_chunk=32 2>&1 | grep -v amdgpu.ids | grep -E "brick order|rep 2" | sort -u
  TILESPMV_X_WINDOW=2 bash scripts/fetch_ab.sh laplacian4096 f64 "-" 2>&1 | grep FETCH
done
unset TILESPMV_BRICK_SHAPE TILESPMV_BRICK_ROWS
echo "== no bricks"
bash scripts/fetch_ab.sh laplacian4096 f64 "-" 2>&1 | grep FETCH
