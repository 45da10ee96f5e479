"""Round 4, VERDICT item 5: the placement lottery of the KKT matrices.  N plans of one Tile_matrix built one after the other in ONE process (all alive at the
same time, as round 3's probe had them), each timed; once without the placement retry (placement_tries=1) and once with the default (3 for plans >= 1 GB).
python scripts/archive/rounds/r4_placement.py [workload] [f32] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
dt = np.float32 if "f32" in sys.argv[2:] else np.float64
N = int([a for a in sys.argv[2:] if a.isdigit()][0]) if [a for a in sys.argv[2:] if a.isdigit()] else 10
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
balg = api.algorithmic_bytes(nnz, rows, n, np.dtype(dt).itemsize)
print("%s (%s) %s, %d rows, %d nnz" % (wl, src, np.dtype(dt).name, rows, nnz), flush=True)
for label, kw in (("no retry (placement_tries=1)", dict(placement_tries=1)), ("default (up to 8 placements for plans >= 1 GB)", dict())):
    plans, out = [], []
    for i in range(N):
        p = api.Plan(tm, rows, n, nnz, **kw)
        t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(3))
        out.append((t, p.info()["placement_tries"]))
        plans.append(p)
        if len(plans) > 3:          # a few instances stay alive, like a solver holding several operators
            plans.pop(0).close()
    for p in plans:
        p.close()
    ts = np.array([t for t, _ in out])
    print("%-44s ms: %s" % (label, " ".join("%.4f" % t for t in ts)))
    print("%-44s placements timed per plan: %s   min %.4f max %.4f spread %.1f %%   B_alg/t %.2f-%.2f of 8 TB/s" % ("", " ".join(str(k) for _, k in out), ts.min(), ts.max(), 100 * (ts.max() / ts.min() - 1),
          balg / ts.max() * 1e-6 / 8000, balg / ts.min() * 1e-6 / 8000), flush=True)
