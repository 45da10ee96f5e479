"""Round 6: absorbed list entries — exactness (host- and device-built plans, SpMV and SpMM) and time, absorb on / off: python scripts/absorb_probe.py wl,wl [f64|f32]"""
import os, sys
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
dt = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
    want = (sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64))
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    for kw in (dict(absorb=0), dict(), dict(desc_dict=0), dict(entry_mode=2), dict(absorb=0, entry_mode=2)):
        for dev in (False, True):
            p = api.Plan.from_csr(rows, n, nnz, rp, ci, v, dtype=dt, deterministic=1, **kw) if dev else api.Plan(tm, rows, n, nnz, deterministic=1, **kw)
            yd.zero_(); p.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
            ok = np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), want)
            ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3))
            i = p.info()
            print("%-14s %s %-28s %s  %.4f ms  list_entries %9d desc %2d stream_bytes %d entry_mode %d %s" % (wl, dt.__name__[5:], kw, "device" if dev else "host  ", ms, i["list_entries"], i["desc_bytes"], i["stream_bytes"], i["entry_mode"], "ok" if ok else "WRONG"), flush=True)
            p.close()
    api.Tile_destroy(tm)
