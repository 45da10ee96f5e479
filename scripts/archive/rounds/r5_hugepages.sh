#!/bin/bash
# Round 5: Tile_create + plan creation of config 4 with and without transparent-huge-page advice on the large host arrays (TILESPMV_HUGEPAGES), three fresh processes each
for hp in 0 1 0 1; do
  for rep in 1 2; do
    TILESPMV_HUGEPAGES=$hp python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tilespmv_amd import api, generators as G
m, n, rp, ci = G.laplacian5pt(4096); nnz = len(ci); v = G.compat_values(nnz)
out = []
for i in range(3):
    t = time.time(); tm = api.Tile_create(m, n, nnz, rp, ci, v); t1 = time.time() - t
    t = time.time(); p = api.Plan(tm, m, n, nnz); t2 = time.time() - t
    out.append((round(t1, 3), round(t2, 3))); p.close(); api.Tile_destroy(tm)
print("TILESPMV_HUGEPAGES=%s (Tile_create s, plan create s) x 3 in one process:" % os.environ["TILESPMV_HUGEPAGES"], out, open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
PY
  done
done
