"""Round 5: SpMM on pooled plans — the native multi-vector kernel (k_pool_mv) against one right-hand side at a time (mv_native = 0), whole Y against scipy."""
import os, sys
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["fem3_68", "fem6_46", "fem3s64_68"]):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in (np.float64, np.float32):
        v = G.compat_values(len(ci), dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        A = sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n))
        tdt = torch.float64 if dt == np.float64 else torch.float32
        xd = torch.from_numpy(G.compat_x(n, dt)).cuda(); yd = torch.zeros(rows + 16, dtype=tdt, device="cuda")
        for nvec in (2, 4, 8):
            X = (np.arange(n * nvec, dtype=np.int64) % 5).astype(dt).reshape(n, nvec)
            want = A @ X.astype(np.float64)
            res = []
            for mvn in (-1, 0):
                p = api.Plan(tm, rows, n, nnz, mv_native=mvn)
                Xd = torch.from_numpy(X).cuda(); Yd = torch.full((rows + 16, nvec), -2.0, dtype=tdt, device="cuda")
                p.spmm(Xd.data_ptr(), Yd.data_ptr(), nvec, st); torch.cuda.synchronize()
                Yh = Yd.cpu().numpy()
                ok = np.array_equal(Yh[:rows].astype(np.float64), want) and (Yh[rows:] == -2.0).all()
                ms = min(p.time_spmm(Xd.data_ptr(), Yd.data_ptr(), nvec, st, warmup=5, reps=20) for _ in range(2))
                t1 = p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=5, reps=20)
                res.append((ms, ok, p.info()["csr_form"]))
                p.close()
            print("%-11s %s nvec %d: native %.4f ms (%s, csr_form %d)  one at a time %.4f ms (%s)  SpMV %.4f ms  -> %.2f x per vector against nvec SpMVs" % (
                wl, dt.__name__[5:], nvec, res[0][0], "ok" if res[0][1] else "WRONG", res[0][2], res[1][0], "ok" if res[1][1] else "WRONG", t1, nvec * t1 / res[0][0]), flush=True)
        api.Tile_destroy(tm)
