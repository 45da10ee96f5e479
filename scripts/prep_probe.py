import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
sys.argv = sys.argv[:1]
import bench
from tilespmv_amd import api, generators as G
for wl, dt in (("powerlaw8000000", np.float64), ("nlpkkt160", np.float32), ("webbase", np.float64), ("laplacian4096", np.float64)):
    m, n, rp, ci, _ = bench.build_matrix(wl)
    rows = (m // 16) * 16; nnz = int(rp[rows])
    vals = G.compat_values(len(ci), dt)
    t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt); t1 = time.time()
    p = api.Plan(tm, rows, n, nnz); t2 = time.time()
    i = p.info()
    print("%-18s Tile_create %.2f s  plan %.2f s (build %.2f, upload %.2f)  device MB %.0f  entry_mode %d" % (wl, t1 - t0, t2 - t1, i["build_us"] * 1e-6, i["upload_us"] * 1e-6, i["device_bytes"] / 1e6, i["entry_mode"]), flush=True)
    p.close(); api.Tile_destroy(tm)
