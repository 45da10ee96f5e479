"""Round 5 (second session): strip cost / XCD window / entry mode on the two cache-resident BASELINE stand-ins."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl, sets in (("scircuit", [dict(), dict(strip_cost=100), dict(strip_cost=200), dict(strip_cost=300), dict(strip_cost=600), dict(strip_cost=200, entry_mode=2), dict(strip_cost=200, entry_mode=1), dict(xcd_chunk=4), dict(xcd_chunk=16), dict(xcd_remap=0),
                               dict(strip_cost=200, xcd_chunk=4), dict(nt_stream=0), dict(desc_dict=0), dict(y_store=0), dict(y_store=1), dict(coo_heavy_min=8), dict(strip_cost=200, csr_split=1), dict(strip_cost=300, csr_split=1)]),
                 ("webbase", [dict(), dict(strip_cost=600), dict(strip_cost=800), dict(strip_cost=1400), dict(strip_cost=2000), dict(xcd_chunk=4), dict(xcd_chunk=16), dict(entry_ordered=1), dict(entry_mode=1), dict(y_store=0), dict(y_store=1), dict(wg_strips=32, entry_mode=2),
                              dict(strip_cost=800, xcd_chunk=4)])):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, hyb=(wl == "scircuit"))
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    for kw in sets:
        p = api.Plan(tm, rows, n, nnz, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 50, 400) for _ in range(4))
        i = p.info()
        print("%-9s %-44s %.5f ms frac %.3f (form %d mode %d ordered %d strip %d tasks %d)" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["entry_ordered"], i["strip_cost"], i["num_tasks"]), flush=True)
        p.close()
    api.Tile_destroy(tm)
