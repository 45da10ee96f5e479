"""Single-GPU estimate of the N-GPU strong-scaling step time: time one rank's shard of the bench workload
(the SpMV needs no collective, so the N-GPU step time is the slowest shard's time)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tilespmv_amd import generators as G
from tilespmv_amd.dist import ShardedSpMV

m, n, rp, ci, src = bench.build_matrix("laplacian4096")
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci)), G.compat_x(n)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
base = None
for world in (1, 2, 4, 8):
    ts = []
    for rank in sorted({0, world // 2, world - 1}):
        sh = ShardedSpMV(rank, world, rows, n, rp, ci, vals)
        t = min(sh.local.time(xd.data_ptr(), yd[sh.r0:].data_ptr(), warmup=10, reps=100) for _ in range(3))
        ts.append(t); sh.close()
    t = max(ts)
    base = base or t
    print("world=%d  slowest shard %.4f ms  -> speedup %.2fx  (%.0f GFLOP/s aggregate)" % (world, t, base / t, 2 * nnz / t * 1e-6), flush=True)
