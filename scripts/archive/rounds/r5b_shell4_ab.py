"""Round 5 (second session): shell4_780 across boxes ran 0.1675-0.1796 ms with the default plan, and single runs of its candidate forms are not comparable between boxes.  One process, every form built once,
timed in turn three times round: default (split, entry mode 2), split with entry mode 0 / strips of 800, pooled (default strips), pooled with full strips."""
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
    forms = [("default", dict()), ("split, entry mode 0, strips 800", dict(csr_split=1, entry_mode=0, strip_cost=800)), ("pooled", dict(csr_split=2)), ("pooled, full strips", dict(csr_split=2, strip_cost=1000000)),
             ("wide pooled", dict(csr_split=3)), ("default again (second plan)", dict())]
    plans = [(lab, api.Plan(tm, rows, n, nnz, placement_tries=1, **kw)) for lab, kw in forms]
    for rnd in range(3):
        for lab, p in plans:
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(2))
            i = p.info()
            print("%-12s round %d  %-34s %.4f ms frac %.3f (form %d mode %d strip %d tasks %d, streams %.0f MB)" % (wl, rnd, lab, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"], i["num_tasks"], i["stream_bytes"] / 1e6), flush=True)
    for _, p in plans: p.close()
    api.Tile_destroy(tm)
