#!/bin/bash
# Round 6, VERDICT item 4 (graph class, stop rule): counters of the default plan of a workload (default rmat22x8) — L1 -> L2 requests, L2 hits / misses, fabric bytes — and its
# kernel times, to set the entry kernels' gather rate beside the chip's ceiling for unshared gathers (scripts/micro/gather_granule.hip: 59.4 G/s from a 64-MB table).
# One rocprofv3 --pmc pass per counter set (no trace flags with --pmc on this pool), one --kernel-trace --stats pass.  Output: gpurun_out/r6_graph_<workload>/
wl=${1:-rmat22x8}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r6_graph_$wl
mkdir -p $out; cd /tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $out/s$i -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check > $out/s$i.log 2>&1 || echo "s$i failed: $set"
  echo "set $i done"
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-check > $out/stats.log 2>&1 || echo "stats failed"
tail -1 $out/stats.log > $out/bench_line.json
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out > $out/summary_print.txt 2>&1
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -12 $out/kernel_stats.csv
cat $out/summary_print.txt | head -60
# the heavy per-dispatch CSVs stay on the box
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*agent_info.csv" -delete
