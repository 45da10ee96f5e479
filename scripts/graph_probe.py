"""Does capturing back-to-back SpMVs of a small matrix into a hipGraph cut the per-launch cost?
python scripts/graph_probe.py [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

wl = sys.argv[1] if len(sys.argv) > 1 else "scircuit"
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci)), G.compat_x(n)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals)
p = api.Plan(tm, rows, n, nnz)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
K = 50
s = torch.cuda.Stream()
def ev_time(f, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    a.record(s)
    for _ in range(reps): f()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
with torch.cuda.stream(s):
    direct = ev_time(lambda: p.spmv_n(xd.data_ptr(), yd.data_ptr(), s.cuda_stream, K)) / K
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        p.spmv_n(xd.data_ptr(), yd.data_ptr(), s.cuda_stream, K)
    graph = ev_time(lambda: g.replay()) / K
print("%s: direct %.2f us / SpMV, hipGraph of %d %.2f us / SpMV" % (wl, direct * 1e3, K, graph * 1e3))
