#!/bin/bash
# round-3: workgroup -> XCD map (windows of 8 x 32 vs round-robin) x resident workgroups per CU (unused-LDS padding)
out=gpurun_out/$1; mkdir -p $out; shift
V='"" TILESPMV_XCD_REMAP=0 TILESPMV_LDS_PAD=6144 TILESPMV_LDS_PAD=12288 TILESPMV_LDS_PAD=20480 TILESPMV_XCD_REMAP=0,TILESPMV_LDS_PAD=6144 TILESPMV_XCD_REMAP=0,TILESPMV_LDS_PAD=12288 TILESPMV_XCD_REMAP=0,TILESPMV_LDS_PAD=20480 TILESPMV_XCD_CHUNK=4 TILESPMV_XCD_CHUNK=4,TILESPMV_LDS_PAD=12288'
for wl in ${@:-nlpkkt160}; do
  eval timeout -k 10 500 python scripts/exp_bench.py $wl $V > $out/$wl${EXP_F64:+_f64}.txt 2>&1
  rc=$?; echo "== $wl ${EXP_F64:+f64} rc=$rc"; grep -v amdgpu.ids $out/$wl${EXP_F64:+_f64}.txt | tail -11
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done
