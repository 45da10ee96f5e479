"""Knob sweep on the cache-resident BASELINE stand-ins (configs 2 / 3): python scripts/small_sweep.py scircuit,webbase"""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v, x = G.compat_values(len(ci), np.float64), G.compat_x(n, np.float64)
    hyb = wl == "scircuit"
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=np.float64, hyb=hyb)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
    import scipy.sparse as sp
    want = sp.csr_matrix((v[:nnz], ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x
    res = []
    for sc, em, cs in itertools.product((0, 50, 100, 200, 400, 800), (-1, 0, 1, 2), (-1, 1, 2, 3)):
        kw = {}
        if sc: kw["strip_cost"] = sc
        if em >= 0: kw["entry_mode"] = em
        if cs >= 0: kw["csr_split"] = cs
        try:
            p = api.Plan(tm, rows, n, nnz, **kw)
        except Exception as e:
            print(wl, kw, "refused", e); continue
        yd.zero_(); p.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
        ok = np.array_equal(yd.cpu().numpy()[:rows], want)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=50, reps=500) for _ in range(3))
        i = p.info()
        res.append((ms, kw, i["num_tasks"], i["entry_mode"], i["csr_form"], i["strip_cost"], ok))
        p.close()
    res.sort(key=lambda t: t[0])
    for r in res[:12] + [t for t in res if not t[1]]:
        print("%-10s %.5f ms  %-50s tasks %6d entry_mode %d csr_form %d strip_cost %d %s" % ((wl, r[0], r[1]) + r[2:6] + ("ok" if r[6] else "WRONG",)), flush=True)
