"""Round 5 (second session): pooled plans — are UNIFORM strips what matters?  r05_pool_strip_cost.txt: fem3_68 gains 4.6 % at a strip cost of 6,400 (every strip holds the maximum of 4 tile-rows) and loses 4 % at
1,600 (a mix of 1- and 2-row strips); fem6_46 loses 7 % at 3,200 / 6,400 (mixes).  Every pooled structure with: the default, every strip full (cost 10^6), and in between; plus the forced pooled form on the shells."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for spec in sys.argv[1].split(","):
    wl, _, form = spec.partition(":")
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v)
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    base = dict(csr_split=int(form)) if form else {}
    for kw in (dict(), dict(strip_cost=6400), dict(strip_cost=12800), dict(strip_cost=1000000), dict(strip_cost=1000000, entry_mode=0), dict(strip_cost=1000000, entry_mode=2)):
        p = api.Plan(tm, rows, n, nnz, placement_tries=1, **base, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        i = p.info()
        print("%-16s %-50s %.4f ms frac %.3f (form %d mode %d strip %d tasks %d = %.2f tile-rows per strip, %d workgroups, split rows %d)" % (spec, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"],
              i["num_tasks"], (rows / 16) / max(1, i["num_tasks"]), (i["num_tasks"] + 15) // 16, i["num_split_rows"]), flush=True)
        p.close()
    api.Tile_destroy(tm)
