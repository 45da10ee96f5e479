#!/bin/bash
# round-3: brick task order (default on large 3-D shards) vs the linear order, XCD window sizes
out=gpurun_out/$1; mkdir -p $out; shift
V='"" TILESPMV_X_WINDOW=0 TILESPMV_XCD_CHUNK=4 TILESPMV_XCD_CHUNK=16 TILESPMV_XCD_CHUNK=32 TILESPMV_XCD_REMAP=0 TILESPMV_XCD_CHUNK=8,TILESPMV_LDS_PAD=12288 TILESPMV_X_WINDOW=2,TILESPMV_STRIP_COST=800'
for wl in ${@:-nlpkkt160}; do
  eval TILESPMV_PLAN_VERBOSE=1 timeout -k 10 500 python scripts/exp_bench.py $wl $V > $out/$wl${EXP_F64:+_f64}.txt 2>&1
  rc=$?; echo "== $wl ${EXP_F64:+f64} rc=$rc"; grep "brick order" $out/$wl${EXP_F64:+_f64}.txt | sort | uniq -c | cut -c1-200; grep -v "amdgpu.ids\|brick order" $out/$wl${EXP_F64:+_f64}.txt | tail -8
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
