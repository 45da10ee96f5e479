#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$1; shift
mkdir -p $out; cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_IFETCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $out/s$i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check "$@" > $out/s$i.log 2>&1 || echo "s$i failed"
  echo "set $i done"
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out | head -60
