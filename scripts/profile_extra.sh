#!/bin/bash
# kernel-trace stats for the other workloads quoted in DESIGN.md §6
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_extra; rm -rf $out; mkdir -p $out; cd /tmp
for wl in nlpkkt160 band40_2000000 scircuit webbase; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $out/$wl.json 2> $out/$wl.err
  cp $(ls $out/$wl/*/*kernel_stats.csv | head -1) $out/${wl}_kernel_stats.csv
  echo "$wl done"; cat $out/$wl.json | cut -c1-400
done
