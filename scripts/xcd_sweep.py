"""Workgroup -> XCD window sweep on the streaming workloads: python scripts/xcd_sweep.py laplacian4096,lap3d256 [f64|f32]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
dt = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    for rep in (1, 2):
        for kw in (dict(), dict(xcd_remap=0), dict(xcd_chunk=4), dict(xcd_chunk=8), dict(xcd_chunk=16), dict(xcd_chunk=32), dict(xcd_chunk=64), dict(xcd_chunk=128), dict(xcd_chunk=256), dict(xcd_chunk=1024), dict(xcd_chunk=4096)):
            p = api.Plan(tm, rows, n, nnz, deterministic=1, **kw)
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3))
            print("%-16s rep %d %-22s %.4f ms" % (wl, rep, kw, ms), flush=True)
            p.close()
    api.Tile_destroy(tm)
