"""Round 5 (second session): Tile_create on the device against the host version, per workload; the device's phases come from TILESPMV_CREATE_VERBOSE."""
import os, sys, time
os.environ["TILESPMV_CREATE_VERBOSE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    dt = np.float32 if wl == "nlpkkt160" else np.float64
    v = G.compat_values(len(ci), dt)
    for rep in range(2):
        t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt); t_h = time.time() - t0
        api.Tile_destroy(tm)
        t0 = time.time(); td = api.Tile_create_device(rows, n, nnz, rp, ci, v, dtype=dt); t_d = time.time() - t0
        api.Tile_destroy(td)
        print("%s rep %d: %d rows, %d nnz: host Tile_create %.3f s, device Tile_create incl. download %.3f s" % (wl, rep, rows, nnz, t_h, t_d), flush=True)
