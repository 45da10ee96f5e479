#!/bin/bash
# Round 6 probe: brick order with 8-row strips on config 4 — time and FETCH_SIZE
export TILESPMV_BRICK_ROWS=8
python3 scripts/knob_time.py laplacian4096 f64 "" x_window=2 x_window=2,xcd_chunk=4 x_window=2,xcd_chunk=16 x_window=2,xcd_chunk=32 2>&1 | grep -v amdgpu.ids
export TILESPMV_X_WINDOW=2
bash scripts/fetch_ab.sh laplacian4096 f64 "-"
