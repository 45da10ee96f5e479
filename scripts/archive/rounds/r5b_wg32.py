import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v)
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    for kw in (dict(), dict(wg_strips=32, entry_mode=2), dict(strip_cost=3200), dict(strip_cost=6400), dict(wg_strips=32, entry_mode=2, strip_cost=3200), dict(wg_strips=32, entry_mode=2, strip_cost=6400), dict(entry_mode=0), dict(entry_mode=0, strip_cost=800)):
        p = api.Plan(tm, rows, n, nnz, placement_tries=1, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        i = p.info()
        print("%-14s %-56s %.4f ms frac %.3f (form %d mode %d strip %d tasks %d wg %d brick %d)" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"], i["num_tasks"], i["wg_strips"], i["brick_order"]), flush=True)
        p.close()
    api.Tile_destroy(tm)
