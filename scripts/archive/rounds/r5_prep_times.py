"""Round 5: where plan creation spends its time (host stages of the unit-stream builder + uploads), per workload.  TILESPMV_PLAN_VERBOSE prints the stage times."""
import os, sys, time
os.environ["TILESPMV_PLAN_VERBOSE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    dt = np.float32 if wl == "nlpkkt160" else np.float64
    v = G.compat_values(len(ci), dt)
    for rep in range(2):
        t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt); t_tc = time.time() - t0
        t0 = time.time(); p = api.Plan(tm, rows, n, nnz, placement_tries=1); t_pc = time.time() - t0
        i = p.info()
        print("%s rep %d: Tile_create %.3f s, plan create %.3f s (build %.3f s, upload %.3f s, %d MB on the device)" % (wl, rep, t_tc, t_pc, i["build_us"] * 1e-6, i["upload_us"] * 1e-6, i["device_bytes"] >> 20), flush=True)
        p.close(); api.Tile_destroy(tm)
