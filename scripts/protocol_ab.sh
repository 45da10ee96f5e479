#!/bin/bash
# The driver's protocol (5 warm-ups + 20 timed steps in a fresh process) under TILESPMV_ABSORB = 0 / 2 / 1, interleaved, same box
for rep in 1 2 3 4; do
for a in 0 2 1; do
  TILESPMV_ABSORB=$a python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('absorb $a rep $rep value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'steady', d['steady_state']['value'], d['steady_state']['kernel_ms'])"
done
done
