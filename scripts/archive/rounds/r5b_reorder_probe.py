"""Round 5 (second session), probe for the next round (no product code): would a symmetric reordering at plan creation pay on the window-shuffled meshes?  Rows sorted by the mean column
index of their nonzeros (k sweeps of that barycentre rule), A' = P A P^T built with scipy, default plan of A' timed beside the default plan of A.  The permutations of x and y a product
would add are priced at 2 x (8 + 4) bytes per row and column at 5 TB/s."""
import os, sys, time
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G

def timed(rows, n, rp, ci):
    nnz = int(rp[rows]); v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    p = api.Plan.from_csr(rows, n, nnz, rp, ci, v)
    ms = min(p.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(3)); i = p.info(); p.close()
    return ms, i

for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    if rows != n and abs(rows - n) > 16: print(wl, "not square"); continue
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    ms0, i0 = timed(rows, n, rp, ci)
    print("%-16s as given:                 %.4f ms frac %.3f (form %d mode %d, streams %.0f MB)" % (wl, ms0, b_alg / ms0 * 1e-6 / 8000, i0["csr_form"], i0["entry_mode"], i0["stream_bytes"] / 1e6), flush=True)
    N = min(rows, n)
    A = sp.csr_matrix((np.ones(nnz, dtype=np.float32), ci[:nnz], rp[:rows + 1]), shape=(rows, n))[:N, :N].tocsr()
    pos = np.arange(N, dtype=np.float64)      # current position of every node
    if os.environ.get("PROBE_RCM"):
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        t0 = time.time()
        S = (A + A.T).tocsr()
        perm = reverse_cuthill_mckee(S, symmetric_mode=True)
        Ap = A[perm][:, perm].tocsr(); Ap.sort_indices()
        t_perm = time.time() - t0
        r2 = (N // 16) * 16
        ms, i = timed(r2, N, Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32))
        extra_ms = 2 * 12.0 * N / 5e12 * 1e3
        print("%-16s reverse Cuthill-McKee:    %.4f ms frac %.3f (form %d mode %d, streams %.0f MB) + ~%.4f ms for permuting x and y -> frac %.3f   [ordering + permuted CSR on the host: %.1f s]" % (
              wl, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["stream_bytes"] / 1e6, extra_ms, b_alg / (ms + extra_ms) * 1e-6 / 8000, t_perm), flush=True)
        continue
    for sweep in range(1, 4):
        t0 = time.time()
        deg = np.maximum(1, np.diff(A.indptr))
        bary = (A @ pos) / deg                  # mean position of a row's neighbours
        order = np.argsort(bary, kind="stable")
        pos = np.empty(N); pos[order] = np.arange(N)
        perm = order                            # new row k = old row perm[k]
        Ap = A[perm][:, perm].tocsr(); Ap.sort_indices()
        t_perm = time.time() - t0
        r2 = (N // 16) * 16
        ms, i = timed(r2, N, Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32))
        extra_ms = 2 * 12.0 * N / 5e12 * 1e3
        print("%-16s reordered, %d sweep(s):     %.4f ms frac %.3f (form %d mode %d, streams %.0f MB) + ~%.4f ms for permuting x and y -> frac %.3f   [ordering + permuted CSR on the host: %.1f s]" % (
              wl, sweep, ms, b_alg / ms * 1e-6 / 8000, i["csr_form"], i["entry_mode"], i["stream_bytes"] / 1e6, extra_ms, b_alg / (ms + extra_ms) * 1e-6 / 8000, t_perm), flush=True)
