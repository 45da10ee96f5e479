import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from tilespmv_amd import api, generators as G
for wl in ["scircuit", "webbase"]:
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, hyb=(wl == "scircuit"))
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    for rep in range(2):
        for kw in (dict(), dict(csr_split=1), dict(csr_split=3), dict(csr_split=3, xcd_chunk=8), dict(csr_split=1, xcd_chunk=8), dict(csr_split=3, entry_mode=2), dict(csr_split=3, entry_mode=0)):
            p = api.Plan(tm, rows, n, nnz, **kw)
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 50, 400) for _ in range(3))
            i = p.info()
            print("%-9s %-40s %.5f ms frac %.3f (form %d mode %d strip %d tasks %d)" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"], i["num_tasks"]), flush=True)
            p.close()
    api.Tile_destroy(tm)
