#!/bin/bash
# round-3: brick order — strip height x XCD window, variants in shuffled positions with duplicates (position effects of exp_bench)
out=gpurun_out/$1; mkdir -p $out; shift
V='TILESPMV_XCD_CHUNK=16 TILESPMV_X_WINDOW=0 TILESPMV_XCD_CHUNK=8 TILESPMV_BRICK_ROWS=4,TILESPMV_XCD_CHUNK=8 TILESPMV_BRICK_ROWS=4,TILESPMV_XCD_CHUNK=16 TILESPMV_BRICK_ROWS=4,TILESPMV_XCD_CHUNK=32 TILESPMV_BRICK_ROWS=2,TILESPMV_XCD_CHUNK=8 TILESPMV_BRICK_ROWS=2,TILESPMV_XCD_CHUNK=16 TILESPMV_X_WINDOW=0,Q=1 TILESPMV_XCD_CHUNK=8,Q=1 TILESPMV_BRICK_ROWS=4,TILESPMV_XCD_CHUNK=8,Q=1 TILESPMV_XCD_CHUNK=16,Q=1'
for wl in ${@:-nlpkkt160}; do
  eval timeout -k 10 600 python scripts/exp_bench.py $wl $V > $out/$wl${EXP_F64:+_f64}.txt 2>&1
  rc=$?; echo "== $wl ${EXP_F64:+f64} rc=$rc"; grep -v "amdgpu.ids" $out/$wl${EXP_F64:+_f64}.txt | tail -12
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
