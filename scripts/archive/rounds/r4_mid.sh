#!/bin/bash
# round 4, mid-round session: whole GPU suite, placement retry on the KKT stand-in, SpMM after the scratch fixes, the full default bench line (how long do the new
# other_workloads take?), the 6-rank rehearsal started bare, counters of the webbase stand-in's gather phase
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4mid; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 600 python scripts/archive/rounds/r4_placement.py nlpkkt160 10 > $out/placement_kkt_f64.txt 2>&1; echo "placement rc=$?"; grep -v amdgpu.ids $out/placement_kkt_f64.txt
timeout -k 10 600 python scripts/archive/rounds/r4_placement.py nlpkkt160 f32 10 > $out/placement_kkt_f32.txt 2>&1; grep -v amdgpu.ids $out/placement_kkt_f32.txt
timeout -k 10 300 python scripts/archive/rounds/r4_placement.py laplacian4096 6 > $out/placement_lap.txt 2>&1; grep -v amdgpu.ids $out/placement_lap.txt
for a in "laplacian4096 f64" "laplacian4096 f32" "nlpkkt160 f32" "nlpkkt160 f64"; do timeout -k 10 300 python scripts/spmm_bench.py $a 1,2,4,8 2>/dev/null | grep "^{" > $out/spmm_$(echo $a | tr ' ' '_').json; cat $out/spmm_$(echo $a | tr ' ' '_').json | cut -c1-600; done
for a in "laplacian4096 f32" "nlpkkt160 f32"; do TILESPMV_LIB_VARIANT=_nd timeout -k 10 300 python scripts/spmm_bench.py $a 8 2>/dev/null | grep "^{" > $out/spmm_nd_$(echo $a | tr ' ' '_').json; echo "no deferred store:"; cat $out/spmm_nd_$(echo $a | tr ' ' '_').json | cut -c1-400; done
( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err ) 2> $out/bench_default.time; echo "bench rc=$?"; tail -3 $out/bench_default.time
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4mid/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "frac_min_bytes", d["roofline"]["frac_min_bytes"], "other_workloads_seconds", d.get("other_workloads_seconds"))
for k, v in d["other_workloads"].items():
    if "error" in v: print(k, v); continue
    for lab in ("coo_in_tile", "default_plan"):
        if lab in v: print("%-16s %-8s %8.5f ms  frac %.3f  min-bytes %.3f  plan-bytes %.3f  check %s  tries %s  %5.1f s" % (k, v["dtype"], v[lab]["ms_per_spmv"], v[lab]["frac_of_8TBps"], v[lab]["frac_min_bytes"], v[lab]["frac_by_plan_bytes"], v[lab]["check"], v[lab]["placement_tries"], v["seconds"]))
PY
( time timeout -k 10 600 python bench.py --gpus 6 --backend gloo --steps 20 --warmup 5 > $out/bench_6ranks.json 2> $out/bench_6ranks.err ) 2> $out/bench_6ranks.time; echo "6 ranks rc=$?"; tail -3 $out/bench_6ranks.time
python -c "
import json;d=json.load(open('gpurun_out/r4mid/bench_6ranks.json'));print(d['value'], d['ranks'], d['devices'], d['check'], {k:(v.get('check_full_y_on_every_rank') or v.get('check_own_rows_on_every_rank')) for k,v in d['with_y_combine'].items()}, d['host_threads_per_rank'], d['usable_host_cores']); print(d['prep_seconds_per_rank'][0], d['prep_seconds_per_rank'][-1])"
scripts/pmc_short.sh r4_webbase --workload webbase > $out/pmc_webbase.log 2>&1; tail -32 $out/pmc_webbase.log
