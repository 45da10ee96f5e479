"""Is the SpMV time data-dependent?  Same matrix structure and plan layout, different value / x contents, interleaved rounds in one
process.  (bench.py's other_workloads show the KKT stand-in in fp32 5-8 % slower with U(-1,1) data than with the reference driver's
i % 10 data; no kernel branches on a value.)  usage: data_dependence_probe.py [workload] [f32|f64]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
    dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
    sys.argv = sys.argv[:1]
    import bench
    m, n, rp, ci, src = bench.build_matrix(wl)
    rows = (m // 16) * 16; nnz = int(rp[rows])
    rng = np.random.default_rng(12345)
    L = len(ci)
    value_sets = {
        "i%10 (reference driver)": lambda: G.compat_values(L, dtype),
        "U(-1,1)": lambda: rng.uniform(-1, 1, L).astype(dtype),
        "random integers 0..9": lambda: rng.integers(0, 10, L).astype(dtype),
        "all 1.0": lambda: np.ones(L, dtype),
        "random sign, 23/52 random mantissa bits, exponent 0": lambda: (1 + rng.random(L)).astype(dtype) * rng.choice(np.array([-1, 1], dtype), L),
    }
    x_sets = {"i%10": G.compat_x(n, dtype), "U(-1,1)": rng.uniform(-1, 1, n).astype(dtype)}
    plans = []
    for vn, mk in value_sets.items():
        vals = mk()
        tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype)
        plans.append((vn, api.Plan(tm, rows, n, nnz)))
        api.Tile_destroy(tm)
        del vals
    yd = torch.zeros(rows + 16, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
    xs = {k: torch.from_numpy(v).cuda() for k, v in x_sets.items()}
    res = {}
    for rnd in range(4):
        for vn, p in plans:
            for xn, xd in xs.items():
                res.setdefault((vn, xn), []).append(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20))
    print("%s %s: ms per SpMV (min of 4 interleaved rounds x 20 launches)" % (wl, np.dtype(dtype).name))
    for (vn, xn), t in res.items():
        print("  values %-58s x %-8s %.4f" % (vn, xn, min(t)), flush=True)

main()
