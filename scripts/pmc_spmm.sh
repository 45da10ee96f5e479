#!/bin/bash
# Memory-pipeline counter sets (as scripts/pmc_short.sh) on the multi-vector product: scripts/pmc_spmm.sh <tag> <workload> <f64|f32> <nvec>
tag=$1; wl=$2; dt=$3; nv=$4
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcspmm_$tag
mkdir -p $out; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d $out/s$i -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py $wl $dt $nv > $out/s$i.log 2>&1 || echo "s$i failed: $set"
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out > $out/summary_print.txt 2>&1
cat $out/summary_print.txt | head -80
