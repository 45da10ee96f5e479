"""Which allocation's placement moves the KKT time (DESIGN S6.13)?  One matrix; P plans with identical contents, X copies of x, Y copies of y
(allocated interleaved with dummy blocks); ms for every (plan, x, y) combination.    python scripts/placement_probe.py [workload] [f64|f32]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tilespmv_amd import api, generators as G
if os.environ.get("PROBE_LIB"):   # a diagnostic build (make VARIANT=... libs)
    os.environ["TILESPMV_LIB_VARIANT"] = os.environ["PROBE_LIB"]
wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, _ = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype)
tdt = torch.float64 if dtype == np.float64 else torch.float32
xs, ys, dummies, plans = [], [], [], []
for i in range(4):
    plans.append(api.Plan(tm, rows, n, nnz))
    dummies.append(torch.empty((37 + 11 * i) << 20, dtype=torch.uint8, device="cuda"))
    xs.append(torch.from_numpy(x).cuda())
    dummies.append(torch.empty((5 + 3 * i) << 20, dtype=torch.uint8, device="cuda"))
    ys.append(torch.zeros(rows + 16, dtype=tdt, device="cuda"))
print("plan x y  ms (min of 4 x 20)")
for pi, p in enumerate(plans):
    for xi, xd in enumerate(xs):
        for yi, yd in enumerate(ys):
            if not (xi == yi or pi == 0):
                continue
            t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(4))
            print("%d %d %d  %.4f   x@%#x y@%#x" % (pi, xi, yi, t, xd.data_ptr(), yd.data_ptr()), flush=True)
