"""Round 5 (second session): wide pooled units (csr_split = 3: windows of 256 columns, byte offsets) against the default plan and the 16-column pooled form, population members, one box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    dt = np.float64
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci), dt); x = G.compat_x(n, dt)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    out = []
    ref = None
    for label, kw in (("default", dict()), ("pooled16", dict(csr_split=2)), ("wide256", dict(csr_split=3)), ("wide256/mode0", dict(csr_split=3, entry_mode=0))):
        try:
            p = api.Plan(tm, rows, n, nnz, placement_tries=1, **kw)
        except Exception as e:
            out.append("%s: %s" % (label, str(e)[:40])); continue
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        y = yd.cpu().numpy()[:rows].copy()
        if ref is None: ref = y
        i = p.info()
        out.append("%s: %.4f ms frac %.3f (form %d mode %d, plan %.0f MB, tasks %d)%s" % (label, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["stream_bytes"] / 1e6, i["num_tasks"], "" if np.array_equal(y, ref) else " Y DIFFERS"))
        p.close()
    print("%s (%d rows, %d nnz): %s" % (wl, rows, nnz, " | ".join(out)), flush=True)
    api.Tile_destroy(tm)
