"""SpMM (nvec 2 / 4 / 8) with absorb 0 / 1 on the stencil workloads, both value types: python scripts/spmm_absorb_ab.py wl,wl"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in (np.float64, np.float32):
        v = G.compat_values(len(ci), dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        for nv in (2, 4, 8):
            X = torch.ones((n, nv), dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda"); Y = torch.zeros((rows + 16, nv), dtype=X.dtype, device="cuda")
            out = []
            for a in (0, 1):
                p = api.Plan(tm, rows, n, nnz, deterministic=1, absorb=a)
                p.reserve_spmm(nv)
                out.append(min(p.time_spmm(X.data_ptr(), Y.data_ptr(), nv, st, warmup=5, reps=30) for _ in range(3)))
                p.close()
            print("%-14s %s nvec %d  absorb 0 %.4f ms  absorb 1 %.4f ms  (%+.1f %%)" % (wl, dt.__name__[5:], nv, out[0], out[1], 100 * (out[1] / out[0] - 1)), flush=True)
        api.Tile_destroy(tm)
