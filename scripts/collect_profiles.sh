#!/bin/bash
# copies the judged summaries of scripts/archive/rounds/r4_final.sh (gpurun_out/prof_<round>_<workload>_<dtype>/) into profiles/ under that round's names: scripts/collect_profiles.sh [r04]
r=${1:-r04}
for d in gpurun_out/prof_${r}_*; do
  [ -d "$d" ] || continue
  tag=${d#gpurun_out/prof_${r}_}
  [ -f $d/kernel_stats.csv ] && cp $d/kernel_stats.csv profiles/${r}_${tag}_kernel_stats.csv
  [ -f $d/bench_under_trace.json ] && grep '^{' $d/bench_under_trace.json | tail -1 > profiles/${r}_${tag}_bench_line_under_trace.json
  for c in FETCH_SIZE WRITE_SIZE; do [ -f $d/pmc_$c.csv ] && grep -E "Kernel_Name|tilespmv" $d/pmc_$c.csv > profiles/${r}_${tag}_pmc_$c.csv; done
  [ -f $d/traffic_${tag}.json ] && cp $d/traffic_${tag}.json profiles/traffic_${tag}.json
done
[ -f gpurun_out/calib_FETCH_SIZE.csv ] && cp gpurun_out/calib_FETCH_SIZE.csv profiles/${r}_calibration_FETCH_SIZE_1GiB_read.csv
ls profiles | grep -c ${r}_
