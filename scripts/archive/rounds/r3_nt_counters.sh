#!/bin/bash
# why the nontemporal streams help: L1 accesses, L1->L2 requests, L2 hits / misses, fabric bytes per launch with the rule off and on
cd $GRAFT_REPO_ROOT
for wl in laplacian4096 powerlaw8000000 nlpkkt160; do
  for nt in 0 1; do
    echo "== $wl nt_stream=$nt"
    extra=""; [ $wl = nlpkkt160 ] && extra="--dtype f64"
    TILESPMV_NT_STREAM=$nt bash scripts/pmc_pair.sh ${wl}_nt$nt --workload $wl $extra
  done
done
