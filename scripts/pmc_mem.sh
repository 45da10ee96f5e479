#!/bin/bash
# Memory-pipeline counters for the unit kernel next to the read-only microbenchmark (reference point).
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$tag
mkdir -p $out; cd /tmp
i=0
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d $out/a$i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check "$@" > $out/a$i.log 2>&1 || echo "a$i failed: $set"
  timeout -k 5 60 rocprofv3 --pmc $set --output-format csv -d $out/b$i -- $GRAFT_REPO_ROOT/scripts/micro/stream_patterns > $out/b$i.log 2>&1 || echo "b$i failed: $set"
  echo "set $i done"
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_units" in k or "k_group_strips<4, 1>" in k or "k_wave_contig<4, 1>" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(json.dumps(res, indent=1))
json.dump(res, open("$out/summary.json", "w"), indent=1)
PY
