"""Multi-vector SpMM vs repeated SpMV on one GPU: python scripts/spmm_bench.py [workload] [dtype] [nvec,nvec,...]
Prints, per nvec, ms per launch, the equivalent per-vector time, and B_alg-based GB/s where
B_alg(nvec) = nnz*(s_v+4) + 4(m+1) + nvec*s_v*(n+m)  (matrix once, X and Y per vector)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tilespmv_amd import api, generators as G  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "laplacian4096"
    dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else np.float64
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    argv, sys.argv = sys.argv, sys.argv[:1]
    import bench
    sys.argv = argv
    m, n, rp, ci, _ = bench.build_matrix(wl)
    nnz = len(ci)
    vals = G.compat_values(nnz, dtype)
    rowA = (m // 16) * 16
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype)
    plan = api.Plan(tp, rowA, n, nnz)
    sv = np.dtype(dtype).itemsize
    out = {"workload": wl, "dtype": np.dtype(dtype).name, "rows": rowA, "nnz": int(rp[rowA]), "results": []}
    rng = np.random.default_rng(0)
    nvs = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (1, 2, 4, 8)
    for nv in nvs:
        X = torch.from_numpy(rng.integers(0, 4, (n, nv)).astype(dtype)).cuda()
        Y = torch.zeros((rowA + 16, nv), dtype=tdt, device="cuda")
        ms = plan.time_spmm(X.data_ptr(), Y.data_ptr(), nv, 0, 10, 50)
        balg = int(rp[rowA]) * (sv + 4) + 4 * (rowA + 1) + nv * sv * (n + rowA)
        out["results"].append({"nvec": nv, "ms": round(ms, 4), "ms_per_vector": round(ms / nv, 4), "GBps_alg": round(balg / ms * 1e-6, 1),
                               "GFLOPs": round(2 * int(rp[rowA]) * nv / ms * 1e-6, 1)})
        del X, Y
    print(json.dumps(out))


if __name__ == "__main__":
    main()
