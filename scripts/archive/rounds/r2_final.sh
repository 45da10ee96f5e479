#!/bin/bash
# Round-2 evidence run on the GPU box: tests, smoke, default bench, other workloads, profiles.  usage: scripts/archive/rounds/r2_final.sh <tag>
tag=${1:-r02}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"
python bench.py --gpus 2 --backend gloo --steps 50 --warmup 10 > $out/bench_line_2ranks_gloo_one_gpu.json 2> $out/bench2.err; echo "bench 2 ranks (gloo, one GPU) rc=$?"
for wl in lap3d256 band40_2000000 powerlaw8000000 powerlaw2000000; do
  timeout -k 10 300 python scripts/exp_bench.py $wl "" 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^ */$wl f64: /"
done > $out/other_workloads.txt
EXP_F64=1 timeout -k 10 300 python scripts/exp_bench.py nlpkkt160 "" 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^ */nlpkkt160 f64: /" >> $out/other_workloads.txt
cat $out/other_workloads.txt
for w in "laplacian4096 f64" "nlpkkt160 f32" "powerlaw8000000 f64" "webbase f64" "scircuit f64" "lap3d256 f64"; do set -- $w; scripts/profile_traffic.sh ${tag}_$1_$2 $1 $2 2>&1 | tail -3; done
scripts/pmc_irregular.sh ${tag}_powerlaw8m --workload powerlaw8000000 > $out/pmcirr_powerlaw8m.log 2>&1
scripts/pmc_irregular.sh ${tag}_webbase --workload webbase > $out/pmcirr_webbase.log 2>&1
scripts/pmc_irregular.sh ${tag}_scircuit --workload scircuit > $out/pmcirr_scircuit.log 2>&1
for wl in "laplacian4096 f64" "laplacian4096 f32" "nlpkkt160 f32"; do timeout -k 10 300 python scripts/spmm_bench.py $wl 2>&1 | grep -v amdgpu.ids | tail -1; done > $out/spmm.jsonl
cat $out/spmm.jsonl | cut -c1-400
