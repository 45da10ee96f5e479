#!/bin/bash
# Produces the judged profile artefacts for one round on the GPU box:
#   gpurun_out/prof_<tag>/kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/prof_<tag>/pmc_*.csv          FETCH_SIZE / WRITE_SIZE passes (separate runs) of the same command
#   gpurun_out/prof_<tag>/calib_*.csv        FETCH_SIZE of a known 1-GiB read in the kernel's own access pattern
# usage: scripts/profile_round.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras "$@" > $out/bench_under_trace.json 2> $out/trace.err
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-check "$@" > /dev/null 2> $out/pmc_$c.err
  cp $(ls $out/pmc_$c/*/*counter_collection.csv | head -1) $out/pmc_$c.csv
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calib -- $GRAFT_REPO_ROOT/scripts/micro/stream_patterns > $out/calib.log 2> $out/calib.err
cp $(ls $out/calib/*/*counter_collection.csv | head -1) $out/calib_FETCH_SIZE.csv
python $GRAFT_REPO_ROOT/scripts/profile_summary.py $out
