#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4mid; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
TILESPMV_PLAN_VERBOSE=1 timeout -k 10 300 python scripts/archive/rounds/r4_placement.py band40_2000000 2 > $out/placement_band40_verbose.txt 2>&1; grep "placement\|ms:" $out/placement_band40_verbose.txt | cut -c1-100 | head -30
TILESPMV_PLAN_VERBOSE=1 timeout -k 10 300 python scripts/archive/rounds/r4_placement.py nlpkkt160 2 > $out/placement_kkt_verbose.txt 2>&1; grep "placement\|ms:" $out/placement_kkt_verbose.txt | cut -c1-100 | head -30
( time timeout -k 10 600 python bench.py --gpus 4 --backend gloo --steps 20 --warmup 5 > $out/bench_4ranks.json 2> $out/bench_4ranks.err ) 2> $out/bench_4ranks.time; echo "4 ranks rc=$?"; tail -3 $out/bench_4ranks.time
python -c "
import json;d=json.load(open('gpurun_out/r4mid/bench_4ranks.json'));print(d['value'], d['ranks'], d['devices'], d['check'], {k:(v.get('check_full_y_on_every_rank') or v.get('check_own_rows_on_every_rank')) for k,v in d['with_y_combine'].items()}, d['host_threads_per_rank'], d['usable_host_cores']); print(d['prep_seconds_per_rank'][0], d['prep_seconds_per_rank'][-1])"
( time timeout -k 10 600 python bench.py --gpus 4 --backend gloo --steps 20 --warmup 5 --workload nlpkkt160 > $out/bench_4ranks_kkt.json 2> $out/bench_4ranks_kkt.err ) 2> $out/bench_4ranks_kkt.time; echo "4 ranks kkt rc=$?"; tail -3 $out/bench_4ranks_kkt.time
python -c "
import json;d=json.load(open('gpurun_out/r4mid/bench_4ranks_kkt.json'));print(d['value'], d['ranks'], d['check'], {k:(v.get('check_full_y_on_every_rank') or v.get('check_own_rows_on_every_rank') or v.get('error')) for k,v in d['with_y_combine'].items()}); print(d['prep_seconds_per_rank'][0])"
scripts/pmc_short.sh r4_webbase --workload webbase > $out/pmc_webbase.log 2>&1; tail -32 $out/pmc_webbase.log
