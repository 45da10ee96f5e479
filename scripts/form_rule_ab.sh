#!/bin/bash
# Round 6: default plans after the pooled form is also considered where a tenth of the shard sits on entry lists (unaligned grids), against TILESPMV_CSR_SPLIT=1 (the classic form, what the rule chose before for these)
for v in "" 1; do
  echo "== TILESPMV_CSR_SPLIT='$v'"
  if [ -n "$v" ]; then export TILESPMV_CSR_SPLIT=$v; else unset TILESPMV_CSR_SPLIT; fi
  python3 scripts/quick_time.py laplacian4090,lap3d250,lap3d200,powerlaw8000000,road3400,bandrand4x3_2000000,uniform8_4000000,rmat22x8,webbase,scircuit,tri2200s4096,plaw18_6000000,nlpkkt160 f64 2>&1 | grep -v amdgpu.ids
done
