"""Round 5 (second session): rmat22x8 — host-built and device-built plans have identical streams and launch forms, entry_ordered = 1, yet their y differ in some bits.  Which plan is not reproducible?
The same plan twice; two host-built plans; two device-built plans; with deterministic=1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "rmat22x8"
m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
v = G.real_values(nnz, np.float64); x = G.real_x(n, nnz, np.float64)
xd = torch.from_numpy(x).cuda()
tm = api.Tile_create(rows, n, nnz, rp, ci, v)
def run(p):
    y = torch.zeros(rows + 16, dtype=torch.float64, device="cuda"); p.spmv(xd.data_ptr(), y.data_ptr()); torch.cuda.synchronize(); return y
for kw in (dict(placement_tries=1), dict(placement_tries=1, x_panel_kb=16384, x_panel_merge=1), dict(placement_tries=1, x_panel_kb=4096, x_panel_merge=2), dict(placement_tries=1, deterministic=1), dict(placement_tries=1, x_panel_kb=0)):
    h1 = api.Plan(tm, rows, n, nnz, **kw); h2 = api.Plan(tm, rows, n, nnz, **kw); d1 = api.Plan.from_csr(rows, n, nnz, rp, ci, v, **kw)
    i = h1.info(); j = d1.info()
    ya, yb, yc, yd = run(h1), run(h1), run(h2), run(d1)
    nd = int((ya != yd).sum())
    print(wl, kw, "same plan twice:", bool(torch.equal(ya, yb)), "| two host plans:", bool(torch.equal(ya, yc)), "| host vs device:", bool(torch.equal(ya, yd)), "(%d rows differ, max rel %.2e)" % (nd, float(((ya - yd).abs() / (ya.abs() + 1e-300)).max())),
          "| facts", {k: (i[k], j[k]) for k in ("entry_mode", "entry_ordered", "num_split_rows", "x_panels", "x_panel_merge", "x_slice_passes", "csr_form", "num_tasks")}, flush=True)
    if nd:
        idx = torch.nonzero(ya != yd).flatten()[:8].cpu().numpy(); print("   first differing rows:", idx.tolist(), "tile-rows", (idx // 16).tolist())
    h1.close(); h2.close(); d1.close()
