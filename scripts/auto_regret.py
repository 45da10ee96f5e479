"""AUTO's regret on the unit form: default plan against csr_split = 1 / 2 / 3 forced, per workload and value type: python scripts/auto_regret.py wl,wl [f64|f32|both]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
dts = {"f64": [np.float64], "f32": [np.float32], "both": [np.float64, np.float32]}[sys.argv[2] if len(sys.argv) > 2 else "f64"]
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in dts:
        v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
        res = {}
        for name, kw in (("auto", {}), ("split", dict(csr_split=1)), ("pooled", dict(csr_split=2)), ("wide", dict(csr_split=3))):
            p = api.Plan(tm, rows, n, nnz, deterministic=1, **kw)
            res[name] = (min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3)), p.info()["csr_form"], p.info()["entry_mode"])
            p.close()
        best = min(res, key=lambda k: res[k][0] if k != "auto" else 1e9)
        regret = res["auto"][0] / res[best][0] - 1
        print("%-20s %s auto form %d mode %d %.4f ms | split %.4f pooled %.4f wide %.4f | best %-6s regret %+5.1f %%%s" % (wl, dt.__name__[5:], res["auto"][1], res["auto"][2], res["auto"][0], res["split"][0], res["pooled"][0], res["wide"][0], best, 100 * regret, "   <<<" if regret > 0.04 else ""), flush=True)
        api.Tile_destroy(tm)
