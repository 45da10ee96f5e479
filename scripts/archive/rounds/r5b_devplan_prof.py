"""Round 5 (second session): only the device preparation (tilespmv_plan_create_from_csr, twice) of one workload — the command rocprofv3 --kernel-trace --stats is pointed at."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
wl = sys.argv[1]
m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
dt = np.float32 if wl == "nlpkkt160" else np.float64
v = G.compat_values(len(ci), dt)
for rep in range(2):
    t0 = time.time(); p = api.Plan.from_csr(rows, n, nnz, rp, ci, v, dtype=dt, placement_tries=1); t = time.time() - t0
    print("%s rep %d: from_csr %.3f s (device Tile_create %.3f s)" % (wl, rep, t, p.info()["tile_create_us"] * 1e-6), flush=True)
    p.close()
