"""Default-plan times of a few workloads, fp64 and fp32 (python scripts/quick_time.py wl,wl [f64|f32|both]); TILESPMV_LIB_VARIANT picks a diagnostic library."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
wls = sys.argv[1].split(",")
dts = {"f64": [np.float64], "f32": [np.float32], "both": [np.float64, np.float32]}[sys.argv[2] if len(sys.argv) > 2 else "both"]
for wl in wls:
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in dts:
        v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        p = api.Plan(tm, rows, n, nnz)
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
        p.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
        import scipy.sparse as sp
        ok = np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64))
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3))
        i = p.info(); isz = np.dtype(dt).itemsize
        print("%-16s %s %.4f ms  frac %.3f  min %.3f  csr_form %d entry_mode %d  %s" % (wl, dt.__name__[5:], ms, api.algorithmic_bytes(nnz, rows, n, isz) / ms * 1e-6 / 8000, isz * (nnz + n + rows) / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], "ok" if ok else "WRONG"), flush=True)
        p.close(); api.Tile_destroy(tm)
