"""Round 5 (second session): shell4_780 (9-point quad mesh x 4 dof, natural order) — the split form leaves 17 % of its nonzeros in entry lists, yet the pooled forms measured slower
in the autotune receipts.  Forms x strip sizes x entry modes."""
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
    for form in (1, 2, 3):
        for em in (0, 2):
            for strip in (400, 800, 1600, 3200):
                p = api.Plan(tm, rows, n, nnz, placement_tries=1, x_panel_kb=0, csr_split=form, entry_mode=em, strip_cost=strip)
                ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
                i = p.info()
                print("%-14s form %d mode %d strip %4d: %.4f ms frac %.3f (tasks %d, streams %.1f MB, desc %d B)" % (wl, i["csr_form"], i["entry_mode"], i["strip_cost"], ms, b_alg / ms * 1e-6 / 8000, i["num_tasks"], i["stream_bytes"] / 1e6, i["desc_bytes"]), flush=True)
                p.close()
    api.Tile_destroy(tm)
