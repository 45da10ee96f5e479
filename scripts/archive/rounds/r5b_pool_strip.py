"""Round 5 (second session): strip cost and split threshold of POOLED plans (their strips hold at most 4 tile-rows): do light tile-rows want full strips, do heavy ones want to stay whole?"""
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
    ref = None
    for kw in (dict(), dict(strip_cost=800), dict(strip_cost=1600), dict(strip_cost=3200), dict(strip_cost=6400), dict(split_above=100000), dict(strip_cost=1600, split_above=100000), dict(strip_cost=3200, split_above=100000)):
        p = api.Plan(tm, rows, n, nnz, placement_tries=1, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        y = yd.cpu().numpy()[:rows].copy()
        if ref is None: ref = y
        i = p.info()
        print("%-14s %-44s %.4f ms frac %.3f (form %d mode %d strip %d tasks %d split rows %d)%s" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"], i["num_tasks"], i["num_split_rows"], "" if np.array_equal(y, ref) else " Y DIFFERS"), flush=True)
        p.close()
    api.Tile_destroy(tm)
