"""Round 5 (second session): grids with fewer 256-thread workgroups than the chip has CUs (the scircuit stand-in: 172) run 128-thread workgroups of 8 strips.  A / B by TILESPMV_SMALL_GRID_WORKGROUPS (0 = off)
in two child processes per workload; also a few small structures of other classes."""
import os, sys, subprocess
import numpy as np
if len(sys.argv) > 2:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    wl = sys.argv[1]; sys.argv = sys.argv[:1]
    import bench
    from tilespmv_amd import api, generators as G
    if wl.startswith("gen:"):
        m, n, rp, ci = eval(wl[4:], {"G": G})
    else:
        m, n, rp, ci, _ = bench.build_matrix(wl)
    rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, hyb=(wl == "scircuit"))
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    import scipy.sparse as sp
    want = sp.csr_matrix((v[:nnz], ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x
    for kw in (dict(), dict(entry_mode=0), dict(entry_mode=1), dict(strip_cost=200), dict(strip_cost=800)):
        p = api.Plan(tm, rows, n, nnz, **kw)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 50, 400) for _ in range(4))
        p.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
        ok = bool(np.array_equal(yd.cpu().numpy()[:rows], want))
        i = p.info()
        print("%-34s small-grid form %-3s %-22s %.5f ms = %.2f us frac %.3f (form %d mode %d strip %d tasks %d -> %d workgroups of 16 strips) %s" % (wl[:34], "off" if os.environ.get("TILESPMV_SMALL_GRID_WORKGROUPS") == "0" else "on", kw, ms, ms * 1e3,
              b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["strip_cost"], i["num_tasks"], (i["num_tasks"] + 15) // 16, "exact" if ok else "WRONG"), flush=True)
        p.close()
    sys.exit(0)
for wl in sys.argv[1].split(";"):
    for sw in ("0", ""):
        env = dict(os.environ)
        if sw: env["TILESPMV_SMALL_GRID_WORKGROUPS"] = sw
        else: env.pop("TILESPMV_SMALL_GRID_WORKGROUPS", None)
        subprocess.run([sys.executable, __file__, wl, "child"], env=env)
