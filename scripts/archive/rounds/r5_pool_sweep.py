"""Round 5: knob sweep of the pooled plans on the FEM class (granularity of the strips: a 27-point x 3 dof tile-row is 81 units = one strip of 1,300 cost units; 59 k strips = 3,685 workgroups = two rounds on 256 CUs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
wls = {"fem3_68": lambda: G.fem_hex(68, 68, 68, 3), "fem6_46": lambda: G.fem_hex(46, 46, 46, 6), "fem3s64_68": lambda: G.fem_hex(68, 68, 68, 3, shuffle=64)}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(wls)
dts = [np.float64, np.float32] if len(sys.argv) <= 2 else [np.float64 if sys.argv[2] == "f64" else np.float32]
for wl in names:
    m, n, rp, ci = wls[wl](); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in dts:
        v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        balg = api.algorithmic_bytes(nnz, rows, n, np.dtype(dt).itemsize)
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
        base = dict(csr_split=2)
        sets = [dict(csr_split=1), dict(), dict(entry_mode=0), dict(entry_mode=2)]
        for sa, sc in ((1200, 400), (800, 400), (800, 200), (600, 200), (400, 200), (400, 100)):
            for em in (0, 2):
                sets.append(dict(entry_mode=em, split_above=sa, split_cap=sa, strip_cost=sc))
        sets += [dict(entry_mode=0, xcd_chunk=8), dict(entry_mode=0, xcd_remap=0), dict(entry_mode=0, nt_stream=0), dict(entry_mode=0, y_store=0), dict(entry_mode=0, x_window=2), dict(entry_mode=0, lds_pad=8192)]
        for kw in sets:
            kw2 = dict(base, **kw) if "csr_split" not in kw else kw
            p = api.Plan(tm, rows, n, nnz, **kw2)
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=5, reps=30) for _ in range(2))
            i = p.info()
            print("%-11s %s %-75s %.4f ms frac %.3f tasks %6d split_rows %6d plan/B_alg %.3f" % (wl, dt.__name__[5:], kw, ms, balg / ms * 1e-6 / 8000, i["num_tasks"], i["num_split_rows"], i["stream_bytes"] / balg), flush=True)
            p.close()
        api.Tile_destroy(tm)
