#!/bin/bash
# L1->L2 requests and fabric bytes of one bench workload under an env variant: scripts/pmc_pair.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcpair_$tag; mkdir -p $out; cd /tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $out/s$i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check "$@" > $out/s$i.log 2>&1 || echo "s$i failed"
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out | grep -v "^{\|^}" | tr -d '\n'; echo
