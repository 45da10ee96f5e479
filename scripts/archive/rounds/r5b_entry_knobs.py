"""Round 5 (second session): knob sweep on entry-dominated shards (tet150s512, power-law 8 M, circuit 4 M): is there a launch form the default rule misses?"""
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
    for kw in (dict(), dict(entry_ordered=0), dict(entry_ordered=1), dict(strip_cost=800), dict(strip_cost=3200), dict(strip_cost=6400), dict(strip_cost=3200, entry_ordered=0), dict(wg_strips=32, entry_mode=2),
               dict(wg_strips=32, entry_mode=2, strip_cost=3200), dict(nt_stream=0), dict(coo_cost=2), dict(coo_cost=8), dict(xcd_chunk=8), dict(xcd_chunk=64), dict(desc_dict=0), dict(y_store=0), dict(y_store=1), dict(entry_mode=0), dict(entry_mode=1), dict(csr_split=1), dict(csr_split=2), dict(csr_split=3), dict(dense_mode=1), dict(dense_mode=2), dict(x_window=2), dict(strip_cost=400), dict(strip_cost=1600)):
        try:
            p = api.Plan(tm, rows, n, nnz, placement_tries=1, x_panel_kb=0, **kw)
        except Exception as e:
            print(wl, kw, "ERR", str(e)[:60]); continue
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        y = yd.cpu().numpy()[:rows].copy()
        if ref is None: ref = y
        i = p.info()
        print("%-18s %-46s %.4f ms frac %.3f (mode %d ordered %d strip %d tasks %d wg %d)%s" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["entry_mode"], i["entry_ordered"], i["strip_cost"], i["num_tasks"], i["wg_strips"], "" if np.array_equal(y, ref) else " Y DIFFERS"), flush=True)
        p.close()
    api.Tile_destroy(tm)
