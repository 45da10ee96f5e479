"""Which placements of the KKT plan are fast?  One plan, placement_tries = 14, TILESPMV_PLAN_VERBOSE prints every placement's block addresses and time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TILESPMV_PLAN_VERBOSE"] = "1"
import torch
from tilespmv_amd import api, generators as G
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, src = bench.build_matrix("nlpkkt160")
rows = (m // 16) * 16; nnz = int(rp[rows])
dt = np.float64
vals = G.compat_values(len(ci), dt)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
for i in range(3):
    print("--- plan", i, flush=True)
    p = api.Plan(tm, rows, n, nnz, placement_tries=14)
    print("info", p.info()["placement_tries"], flush=True)
    p.close()
