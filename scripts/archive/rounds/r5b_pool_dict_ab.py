"""Round 5 (second session): pooled plans with 20-byte descriptors (desc_dict=0) against 8-byte descriptors + pattern dictionary (default), same box, same session."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    dt = np.float32 if wl.endswith(":f32") else np.float64
    name = wl.split(":")[0]
    m, n, rp, ci, _ = bench.build_matrix(name); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci), dt); x = G.compat_x(n, dt)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    b_alg = api.algorithmic_bytes(nnz, rows, n, np.dtype(dt).itemsize)
    out = []
    for rep in range(2):
        for dd in (0, -1):
            p = api.Plan(tm, rows, n, nnz, desc_dict=dd, placement_tries=1)
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 50) for _ in range(3))
            i = p.info()
            out.append("desc %2d B: %.4f ms (frac %.3f, plan %.1f MB)" % (i["desc_bytes"], ms, b_alg / ms * 1e-6 / 8000, i["stream_bytes"] / 1e6))
            p.close()
    print("%s %s: %s" % (name, np.dtype(dt).name, " | ".join(out)), flush=True)
    api.Tile_destroy(tm)
