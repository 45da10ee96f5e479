"""Do the slow and the fast instances of one plan differ in address-translation counters?  N identical plans of one workload; each instance runs
R SpMVs in a row (instance 0 first), so the dispatch order of a rocprofv3 pass maps back to instances.
    rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum -- python scripts/instance_counters.py nlpkkt160 f64
Without a profiler it prints the per-instance times (hip events)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
R = int(sys.argv[4]) if len(sys.argv) > 4 else 10
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, _ = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
plans = [api.Plan(tm, rows, n, nnz) for _ in range(N)]
for p in plans:
    for _ in range(3): p.spmv(xd.data_ptr(), yd.data_ptr())
torch.cuda.synchronize()
for i, p in enumerate(plans):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R): p.spmv(xd.data_ptr(), yd.data_ptr())
    b.record(); torch.cuda.synchronize()
    print("instance %d: %.4f ms per SpMV (%d in a row)" % (i, a.elapsed_time(b) / R, R), flush=True)
