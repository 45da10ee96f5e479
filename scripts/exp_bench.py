"""Experiment driver: time plan variants on one workload in ONE process (interleaved rounds)."""
import os, sys, time, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "laplacian4096"
    variants = sys.argv[2:] or ["TILESPMV_STRIP_COST=192"]
    sys.argv = sys.argv[:1]
    import bench
    m, n, rp, ci, src = bench.build_matrix(wl)
    dtype = np.float32 if ((wl == "nlpkkt160" and not os.environ.get("EXP_F64")) or os.environ.get("EXP_F32")) else np.float64
    rows = (m // 16) * 16; nnz = int(rp[rows])
    vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
    tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    import scipy.sparse as sp
    seg = sp.csr_matrix((vals[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64)
    balg = api.algorithmic_bytes(nnz, rows, n, np.dtype(dtype).itemsize)
    plans = []
    for v in variants:
        kv = dict(s.split("=") for s in v.split(",") if s)
        libv = kv.pop("LIB", None)   # LIB=_abl1: a diagnostic build made with `make VARIANT=_abl1 ... libs` (same Tile_matrix struct)
        for k, val in kv.items(): os.environ[k] = val
        base_lib = tm._lib
        if libv is not None:
            from tilespmv_amd import _lib
            os.environ["TILESPMV_LIB_VARIANT"] = libv; _lib._CACHE.pop(np.dtype(dtype), None)
            tm._lib = _lib.load(dtype)
            os.environ.pop("TILESPMV_LIB_VARIANT"); _lib._CACHE[np.dtype(dtype)] = base_lib
        p = api.Plan(tm, rows, n, nnz)
        tm._lib = base_lib
        for k in kv: os.environ.pop(k)
        yd.fill_(-1); p.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
        ok = bool(np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), seg)) if seg is not None else None
        plans.append((v, p, ok))
    res = {v: [] for v, _, _ in plans}
    for rnd in range(5):
        for v, p, ok in plans:
            res[v].append(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20))
    for v, p, ok in plans:
        t = np.array(res[v]); i = p.info()
        print("%-60s ok=%s min %.4f med %.4f ms  %.0f GB/s alg (%.1f%%)  stream %.0f GB/s  tasks=%d kernel=%d" % (
            v, ok, t.min(), np.median(t), balg / t.min() * 1e-6, balg / t.min() * 1e-6 / 80, i["stream_bytes"] / t.min() * 1e-6, i["num_tasks"], i["kernel"]), flush=True)

main()
