#!/bin/bash
# Kernel durations + HBM traffic per launch for one bench workload, as the judged artefacts want them:
#   gpurun_out/prof_<tag>/kernel_stats.csv            rocprofv3 --kernel-trace --stats of `bench.py --workload W`
#   gpurun_out/prof_<tag>/bench_under_trace.json      the bench line printed by that same traced run
#   gpurun_out/prof_<tag>/pmc_{FETCH,WRITE}_SIZE.csv  separate --pmc passes of the same command
#   gpurun_out/prof_<tag>/traffic_<W>_<dtype>.json    bytes per launch of the dominant kernel (FETCH_SIZE x calibration + WRITE_SIZE)
# usage: scripts/profile_traffic.sh <tag> <workload> <f64|f32> [calib_dir]   (calibration: scripts/profile_round.sh output, optional)
tag=$1; wl=$2; dt=$3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out; cd /tmp
args="--workload $wl --dtype $dt --no-cpu-baseline --no-extras"
# plans that choose something by timing at creation (column panels per pass) must be THE SAME plan in all three passes, and their calibration launches must not be in the
# profiles: one untraced run decides, the profiled runs get the choice as a knob
if [ -z "$TILESPMV_X_PANEL_MERGE" ] && [ -z "$TILESPMV_X_SLICE_PASSES" ]; then
  read m sp < <(python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-check $args 2>/dev/null | python -c "import sys,json; c=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['config']; print(c['x_panel_merge'], c['x_slice_passes'])" 2>/dev/null)
  export TILESPMV_X_PANEL_MERGE=${m:-0} TILESPMV_X_SLICE_PASSES=${sp:-0}
  echo "entry-list form fixed for the profiled runs: panels per pass $TILESPMV_X_PANEL_MERGE, slice passes $TILESPMV_X_SLICE_PASSES"
fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 $args > $out/bench_under_trace.json 2> $out/trace.err || echo "trace failed"
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-check $args > /dev/null 2> $out/pmc_$c.err || echo "$c failed"
  cp $(ls $out/pmc_$c/*/*counter_collection.csv | head -1) $out/pmc_$c.csv
done
if [ ! -f $GRAFT_REPO_ROOT/gpurun_out/calib_FETCH_SIZE.csv ]; then
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/calib -- $GRAFT_REPO_ROOT/scripts/micro/stream_patterns > $out/calib.log 2> $out/calib.err
  cp $(ls $out/calib/*/*counter_collection.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/calib_FETCH_SIZE.csv
fi
python $GRAFT_REPO_ROOT/scripts/traffic_json.py $out $wl $dt $GRAFT_REPO_ROOT/gpurun_out/calib_FETCH_SIZE.csv
