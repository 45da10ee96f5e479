import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
sys.argv = sys.argv[:2]
import bench
from tilespmv_amd import api, generators as G, _lib
wl = sys.argv[1]
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), np.float64), G.compat_x(n, np.float64)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
for lib in ("", "_old"):
    if lib:
        os.environ["TILESPMV_LIB_VARIANT"] = lib; _lib._CACHE.pop(np.dtype(np.float64), None)
    tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=np.float64)
    for kw in (dict(), dict(x_slice_passes=0, x_panel_merge=0), dict(x_slice_passes=1), dict(x_slice_passes=2)):
        p = api.Plan(tm, rows, n, nnz, **kw)
        t = p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20)
        i = p.info()
        print(wl, lib or "new", kw, "ms %.4f" % t, {k: i[k] for k in ("entry_mode", "x_panels", "x_panel_merge", "x_slice_passes", "stream_bytes", "timed_choices_us", "entry_ordered")}, flush=True)
        p.close()
