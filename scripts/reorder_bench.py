"""Permuted-numbering plans (round 6): kernel-only time of the default plan in the natural numbering and after reverse Cuthill-McKee, what the two vector permutations cost,
and the product's time amortised over K products between them.  python scripts/reorder_bench.py [workload ...]  -> one JSON record per workload (also used by bench.py)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKLOADS = ["tet150s512", "tri2200s4096", "fem3s64_68"]


def measure(wl, torch, api, G, build_matrix, dtype=np.float64, reps=40):
    import scipy.sparse as sp
    m, n, rp, ci, src = build_matrix(wl)
    rows = (m // 16) * 16
    A = sp.csr_matrix((np.ones(int(rp[rows]), dtype=np.int8), ci[:int(rp[rows])], rp[:rows + 1]), shape=(rows, n))[:, :rows].tocsr()   # square leading block
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    del A
    n = rows; nnz = len(ci)
    vals, x = G.compat_values(nnz, dtype), G.compat_x(n, dtype)
    t0 = time.time(); perm = api.reorder_rcm(n, rp, ci); t_rcm = time.time() - t0
    t0 = time.time(); brp, bci, bv = api.csr_permute(n, rp, ci, vals, perm, dtype=dtype); t_perm_csr = time.time() - t0
    bw = (api.csr_bandwidth(n, rp, ci), api.csr_bandwidth(n, brp, bci))
    st = torch.cuda.current_stream().cuda_stream
    xd = torch.from_numpy(x).cuda(); pd = torch.from_numpy(perm).cuda()
    xp = torch.zeros(n + 16, dtype=xd.dtype, device="cuda"); y0 = torch.zeros(n + 16, dtype=xd.dtype, device="cuda"); yp = torch.zeros_like(y0); yo = torch.zeros_like(y0)
    b_alg = api.algorithmic_bytes(nnz, n, n, np.dtype(dtype).itemsize)
    out = {"workload": wl, "source": src, "rows": n, "nnz": nnz, "bandwidth_natural": bw[0], "bandwidth_rcm": bw[1], "rcm_seconds": round(t_rcm, 2), "csr_permute_seconds": round(t_perm_csr, 2)}
    p0 = api.Plan.from_csr(n, n, nnz, rp, ci, vals, dtype=dtype)
    t_nat = p0.time(xd.data_ptr(), y0.data_ptr(), st, warmup=10, reps=reps)
    i0 = p0.info(); p0.close()
    p1 = api.Plan.from_csr(n, n, nnz, brp, bci, bv, dtype=dtype)
    api.permute_vector(xd.data_ptr(), xp.data_ptr(), pd.data_ptr(), n, scatter=False, stream=st, dtype=dtype)
    t_rcm_k = p1.time(xp.data_ptr(), yp.data_ptr(), st, warmup=10, reps=reps)
    api.permute_vector(yp.data_ptr(), yo.data_ptr(), pd.data_ptr(), n, scatter=True, stream=st, dtype=dtype)
    torch.cuda.synchronize()
    ok = bool(np.array_equal(yo.cpu().numpy()[:n], y0.cpu().numpy()[:n]))
    i1 = p1.info(); p1.close()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        api.permute_vector(xd.data_ptr(), xp.data_ptr(), pd.data_ptr(), n, scatter=False, stream=st, dtype=dtype)
        api.permute_vector(yp.data_ptr(), yo.data_ptr(), pd.data_ptr(), n, scatter=True, stream=st, dtype=dtype)
    e0.record()
    for _ in range(20):
        api.permute_vector(xd.data_ptr(), xp.data_ptr(), pd.data_ptr(), n, scatter=False, stream=st, dtype=dtype)
        api.permute_vector(yp.data_ptr(), yo.data_ptr(), pd.data_ptr(), n, scatter=True, stream=st, dtype=dtype)
    e1.record(); torch.cuda.synchronize()
    t_pair = e0.elapsed_time(e1) / 20
    frac = lambda t: round(b_alg / t * 1e-6 / 8000.0, 4)
    out.update({"natural": {"ms_per_spmv": round(t_nat, 5), "frac": frac(t_nat), "csr_form": i0["csr_form"], "entry_mode": i0["entry_mode"], "stream_bytes": i0["stream_bytes"]},
                "rcm": {"ms_per_spmv": round(t_rcm_k, 5), "frac": frac(t_rcm_k), "csr_form": i1["csr_form"], "entry_mode": i1["entry_mode"], "stream_bytes": i1["stream_bytes"]},
                "kernel_speedup": round(t_nat / t_rcm_k, 3), "permute_x_plus_unpermute_y_ms": round(t_pair, 5),
                "amortised_over_K_products": {str(K): {"ms_per_product": round(t_rcm_k + t_pair / K, 5), "frac": frac(t_rcm_k + t_pair / K), "vs_natural": round(t_nat / (t_rcm_k + t_pair / K), 3)} for K in (1, 10, 100)},
                "check_unpermuted_y_equals_natural_y_exact": "pass" if ok else "FAIL"})
    return out


if __name__ == "__main__":
    import torch
    wls = sys.argv[1:] or WORKLOADS
    sys.argv = sys.argv[:1]
    import bench
    from tilespmv_amd import api, generators as G
    for wl in wls:
        print(json.dumps(measure(wl, torch, api, G, bench.build_matrix)), flush=True)
