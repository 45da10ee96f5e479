"""Round 5 (second session): shards whose cost is mostly units but which still have >= 5 list entries per tile-row (the rule that switches the workgroup entry mode on):
does entry mode 0 (entries travel with the strips) with shorter strips run faster there?  The autotune receipts said so for shell4_780 (0.178 -> 0.160 ms)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    f32 = wl.endswith(":f32"); wl = wl.split(":")[0]
    dt = np.float32 if f32 else np.float64
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci), dt); x = G.compat_x(n, dt)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    b_alg = api.algorithmic_bytes(nnz, rows, n, 4 if f32 else 8)
    for kw in (dict(), dict(entry_mode=0), dict(entry_mode=0, strip_cost=400), dict(entry_mode=0, strip_cost=800), dict(entry_mode=0, strip_cost=1600), dict(entry_mode=2, strip_cost=800), dict(entry_mode=2, strip_cost=400),
               dict(entry_mode=1, strip_cost=800)):
        p = api.Plan(tm, rows, n, nnz, placement_tries=1, x_panel_kb=0, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3))
        i = p.info()
        print("%-16s %-40s %.4f ms frac %.3f (form %d mode %d strip %d tasks %d; streams %.1f MB)" % (wl + (" f32" if f32 else ""), kw, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"],
              i["num_tasks"], i["stream_bytes"] / 1e6), flush=True)
        p.close()
    api.Tile_destroy(tm)
