#!/bin/bash
# Round 6: the chunked entry lists (10 / 12 B per entry) against round 5's 12-byte records (LIB=_old = the commit before), and the trip / occupancy variants of the new
# reader (LIB=_w6: ECOO2_MIN_WAVES 6, _ct2 / _ct4: WCOO_HEAVY_CT), one process per workload (scripts/exp_bench.py).  Run through gpurun; results: gpurun_out/r6_ab_entries/.
out=gpurun_out/r6_ab_entries; mkdir -p $out
for spec in ${@:-powerlaw8000000 webbase scircuit circuit4000000 bandrand4x3_2000000 uniform8_4000000 rmat22x8 tri2200s4096 nlpkkt160:f64 nlpkkt160:f32 webbase:f32 powerlaw8000000:f32}; do
  wl=${spec%%:*}; dt=${spec#*:}; [ "$dt" = "$spec" ] && dt=f64
  if [ $dt = f32 ]; then export EXP_F32=1; unset EXP_F64; else export EXP_F64=1; unset EXP_F32; fi
  echo "== $wl $dt"
  timeout -k 10 400 python scripts/exp_bench.py $wl "Q=1" "LIB=_old,Q=1" "LIB=_ct2,Q=1" "LIB=_ct4,Q=1" 2>&1 | grep -v amdgpu.ids | tee $out/${wl}_$dt.txt
done
