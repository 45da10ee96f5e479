"""Does every plan keep the promise its facts make?  For the 26 structures of the population (scripts/population_sweep.py), REAL-valued data, fp64 and fp32:
the default plan and the deterministic = 1 plan, each launched three times, plus a second plan of the same options and the device-built plan of the same options.
A plan whose facts say TILESPMV_INFO_ENTRY_ORDERED = 1 (and no column slices) must give the same bits every time and across builds of the same launch form; one whose facts say 0 may differ
(reported, not failed).  SpMM nvec 2 twice on the same plan as well.     python scripts/reproducibility_sweep.py [keys]      exit code 1 on a broken promise"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
keys = sys.argv[1].split(",") if len(sys.argv) > 1 else None
sys.argv = sys.argv[:1]
import bench
from population_sweep import POPULATION
from tilespmv_amd import api, generators as G

def run(p, xd, rows, dt):
    y = torch.full((rows + 16,), 5.0, dtype=dt, device="cuda"); p.spmv(xd.data_ptr(), y.data_ptr()); torch.cuda.synchronize(); return y[:rows].clone()

broken = 0; t_all = time.time()
for key, wl, klass in POPULATION:
    if keys and key not in keys: continue
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dtype, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        if dtype == np.float32 and key not in ("nlpkkt160_f64", "rmat22x8", "fem3_68", "powerlaw8m", "shell4_780", "circuit4m"): continue   # (fp32: a subset — the builders are the same code)
        v, x = G.real_values(nnz, dtype), G.real_x(n, nnz, dtype)
        xd = torch.from_numpy(x).cuda()
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dtype)
        for kw in (dict(), dict(deterministic=1)):
            p1 = api.Plan(tm, rows, n, nnz, **kw); p2 = api.Plan(tm, rows, n, nnz, **kw); pd = api.Plan.from_csr(rows, n, nnz, rp, ci, v, dtype=dtype, **kw)
            i1, i2, i3 = p1.info(), p2.info(), pd.info()
            promised = bool(i1["entry_ordered"]) and i1["x_slice_passes"] == 0
            form = lambda i: (i["entry_mode"], i["entry_ordered"], i["x_panels"], i["x_panel_merge"], i["x_slice_passes"], i["csr_form"], i["num_tasks"])
            ya, yb, yc = run(p1, xd, rows, tdt), run(p1, xd, rows, tdt), run(p1, xd, rows, tdt)
            same_launches = bool(torch.equal(ya, yb) and torch.equal(ya, yc))
            y2 = run(p2, xd, rows, tdt); y3 = run(pd, xd, rows, tdt)
            same_build = bool(torch.equal(ya, y2)) if form(i1) == form(i2) and bool(i2["entry_ordered"]) else None
            same_dev = bool(torch.equal(ya, y3)) if form(i1) == form(i3) and bool(i3["entry_ordered"]) else None
            mv = None
            if not kw:   # SpMM nvec 2 twice
                X = np.ascontiguousarray(np.stack([x, x[::-1]], axis=1)); Xd = torch.from_numpy(X).cuda()
                Ys = []
                for _ in range(2):
                    Yd = torch.zeros((rows + 16, 2), dtype=tdt, device="cuda"); p1.spmm(Xd.data_ptr(), Yd.data_ptr(), 2); torch.cuda.synchronize(); Ys.append(Yd[:rows].clone())
                mv = bool(torch.equal(Ys[0], Ys[1]))
                del Xd, Ys, Yd
            bad = promised and (not same_launches or same_build is False or same_dev is False or mv is False)
            broken += bad
            print("%-16s %s %-20s facts: ordered %d slices %d panels %d/%d split rows %d form %d | three launches identical: %s | second plan: %s | device-built plan: %s | SpMM nvec 2 twice: %s%s" % (
                  key, np.dtype(dtype).name, kw or "{default}", i1["entry_ordered"], i1["x_slice_passes"], i1["x_panels"], i1["x_panel_merge"], i1["num_split_rows"], i1["csr_form"], same_launches,
                  same_build if same_build is not None else "other launch form", same_dev if same_dev is not None else "other launch form", mv, "   <-- BROKEN PROMISE" if bad else ("" if promised else "   (no promise made)")), flush=True)
            p1.close(); p2.close(); pd.close()
        api.Tile_destroy(tm); del xd
print("REPRODUCIBILITY SWEEP: %d broken promises, %.0f s" % (broken, time.time() - t_all))
sys.exit(1 if broken else 0)
