#!/bin/bash
# round-3: x windows (brick task order + LDS-staged x segments) on the stencil-like workloads
out=gpurun_out/$1; mkdir -p $out; shift
V='"" TILESPMV_X_WINDOW=1 TILESPMV_X_WINDOW=1,TILESPMV_STRIP_COST=800 TILESPMV_X_WINDOW=1,TILESPMV_STRIP_COST=400 TILESPMV_X_WINDOW=1,TILESPMV_XCD_CHUNK=8 TILESPMV_X_WINDOW=1,TILESPMV_XCD_REMAP=0 TILESPMV_X_WINDOW=1,TILESPMV_WAVE_COO=0'
for wl in ${@:-nlpkkt160}; do
  eval TILESPMV_PLAN_VERBOSE=1 timeout -k 10 500 python scripts/exp_bench.py $wl $V > $out/$wl${EXP_F64:+_f64}.txt 2>&1
  rc=$?; echo "== $wl rc=$rc"; grep -v amdgpu.ids $out/$wl${EXP_F64:+_f64}.txt | tail -16
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
