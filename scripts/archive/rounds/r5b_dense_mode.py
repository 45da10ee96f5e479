"""Round 5 (second session): dense tiles of the FEM-6 class on the matrix cores (their own pass) against as pooled units in the unit kernel."""
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
    for kw in (dict(), dict(dense_mode=1), dict(dense_mode=2), dict(dense_mode=2, entry_mode=0), dict(dense_mode=1, entry_mode=0)):
        p = api.Plan(tm, rows, n, nnz, placement_tries=1, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        y = yd.cpu().numpy()[:rows].copy()
        if ref is None: ref = y
        i = p.info()
        print("%-14s %-36s %.4f ms frac %.3f (form %d dense_mode %d entry mode %d, plan %.0f MB, tasks %d)%s" % (wl, kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["dense_mode"], i["entry_mode"], i["stream_bytes"] / 1e6, i["num_tasks"], "" if np.array_equal(y, ref) else " Y DIFFERS"), flush=True)
        p.close()
    api.Tile_destroy(tm)
