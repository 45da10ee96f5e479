"""Quick GPU shake-out (manual tool, lives under tests/ because it uses the oracle): parity of the HIP path
vs the oracle on all small matrices x kernels x modes + a first timing.  Run: python tests/gpu_shakeout.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases as tests_cases
sys.modules['tests_cases']=tests_cases
from oracle.oracle import CpuImpl
from tilespmv_amd import generators as G, api

def run_case(name, m, n, rp, ci, dt, hyb, coo_mode, dense_mode, real=False, kernel=0):
    nnz = len(ci)
    rowA = (m // 16) * 16
    if real:
        rng = np.random.default_rng(12345)
        vals = rng.uniform(-1, 1, nnz).astype(dt); x = rng.uniform(-1, 1, n).astype(dt)
    else:
        vals = G.compat_values(nnz, dt); x = G.compat_x(n, dt)
    O = CpuImpl("oracle", dt)
    to = O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb)
    so = O.spmv(to, rowA, n, nnz, rp, ci, vals, x)
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dt, hyb=hyb)
    plan = api.Plan(tp, rowA, n, nnz, coo_mode=coo_mode, dense_mode=dense_mode, kernel=kernel)
    xd = torch.from_numpy(x).cuda(); yd = torch.full((rowA + 16,), 777.0, dtype=xd.dtype, device="cuda")
    plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
    y = yd.cpu().numpy()[:rowA]
    if real:
        absax = np.zeros(rowA); np.add.at(absax, np.repeat(np.arange(m), np.diff(rp))[:rp[rowA]], np.abs(vals[:rp[rowA]].astype(np.float64) * x[ci[:rp[rowA]]].astype(np.float64)))
        tol = (1e-12 if dt == np.float64 else 1e-5) * absax + 1e-300
        bad = int(np.count_nonzero(np.abs(y.astype(np.float64) - so["y_golden"].astype(np.float64)) > tol))
    else:
        bad = int(np.count_nonzero(y != so["y"]))
    info = plan.info()
    print("%-12s %-7s k=%d hyb=%d coo=%d dns=%d real=%d rows=%d nnz=%d tasks=%d split=%d fb=%d -> mismatches %d" % (
        name, np.dtype(dt).name, info["kernel"], hyb, info["coo_mode"], info["dense_mode"], real, rowA, nnz, info["num_tasks"], info["num_split_rows"], info["fallback_nnz"], bad), flush=True)
    if bad:
        w = np.nonzero(y != so["y"])[0][:8]
        print("   first bad rows", w, y[w], so["y"][w])
    plan.close()
    return bad

total = 0
from tests_cases import SMALL
cases = [(k, g()) for k, g in SMALL.items()] + [("lap64", G.laplacian5pt(64)), ("band_8", G.band(4096, 8)), ("band_40", G.band(4096, 40)), ("band1000", G.band(1000, 3)),
         ("allfmt", G.all_formats()), ("allfmt_pad", G.all_formats(cols_pad=5)), ("rand", G.random_uniform(500, 700, 0.02, 3)),
         ("pl", G.powerlaw(20000)), ("circ", G.circuit_like(8000))]
for dt in (np.float64, np.float32):
    for name, (m, n, rp, ci) in cases:
        for hyb in (False, True):
            for coo in (1, 2):
                for dns in (1, 2):
                    for kern in (1, 2):
                        total += run_case(name, m, n, rp, ci, dt, hyb, coo, dns, kernel=kern)
        total += run_case(name, m, n, rp, ci, dt, True, 0, 0, real=True, kernel=2)
print("TOTAL MISMATCHES", total, flush=True)

# first timing
for N in (1024,):
    m, n, rp, ci = G.laplacian5pt(N); nnz = len(ci)
    vals = G.compat_values(nnz); x = G.compat_x(n)
    t0 = time.time(); tp = api.Tile_create(m, n, nnz, rp, ci, vals); t1 = time.time()
    plan = api.Plan(tp, m, n, nnz); t2 = time.time()
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(m + 16, dtype=torch.float64, device="cuda")
    ms = plan.time(xd.data_ptr(), yd.data_ptr(), warmup=5, reps=20)
    balg = api.algorithmic_bytes(nnz, m, n, 8)
    print("laplacian %d^2: Tile_create %.2fs plan %.2fs  spmv %.4f ms  %.1f GFLOP/s  %.1f GB/s alg (%.1f%% of 8TB/s) info=%s" % (
        N, t1 - t0, t2 - t1, ms, 2 * nnz / ms * 1e-6, balg / ms * 1e-6, balg / ms * 1e-6 / 80, plan.info()), flush=True)
    if N == 1024:
        O = CpuImpl("oracle"); to = O.tile_create(m, n, nnz, rp, ci, vals); so = O.spmv(to, m, n, nnz, rp, ci, vals, x)
        print("  parity 1024^2:", int(np.count_nonzero(yd.cpu().numpy()[:m] != so["y"])))
    plan.close()
