"""Stress the in-kernel split-row fix-up (last finishing piece sums the slots): rows with hundreds of pieces spread over
many workgroups / XCDs, thousands of launches, result compared on the device after every launch.
python scripts/fix_inline_stress.py [launches]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(0)
rows, cols = 256, 400000
R, C = [], []
for r in (3, 17, 40, 41, 100, 200, 255):                      # long rows: ~60k..300k entries each
    k = int(rng.integers(60000, 300000)); R.append(np.full(k, r)); C.append(rng.choice(cols, k, replace=False))
for r in range(rows):                                         # plus a sprinkle everywhere
    k = int(rng.integers(0, 40)); R.append(np.full(k, r)); C.append(rng.choice(cols, k, replace=False))
r = np.concatenate(R); c = np.concatenate(C)
key = np.unique(r.astype(np.int64) * cols + c); r, c = key // cols, key % cols
m, n, rp, ci = G.from_coo(rows, cols, r, c)
nnz = len(ci)
bad_total = 0
for dtype, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    vals = rng.integers(1, 4, nnz).astype(dtype); x = rng.integers(0, 4, n).astype(dtype)
    import scipy.sparse as sp
    want = torch.from_numpy((sp.csr_matrix((vals.astype(np.float64), ci, rp), shape=(m, n)) @ x.astype(np.float64)).astype(dtype)).cuda()
    tm = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dtype)
    for env in ({}, {"TILESPMV_XCD_REMAP": "0"}, {"TILESPMV_XCD_REMAP": "0", "TILESPMV_SPLIT_ABOVE": "300", "TILESPMV_STRIP_COST": "48"}):
        os.environ.update(env)
        p = api.Plan(tm, m, n, nnz)
        for k in env: os.environ.pop(k)
        info = p.info()
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(m + 16, dtype=tdt, device="cuda")
        bad = 0
        for it in range(launches):
            p.spmv(xd.data_ptr(), yd.data_ptr())
            if it % 8 == 0:
                p.spmv(xd.data_ptr(), yd.data_ptr()); p.spmv(xd.data_ptr(), yd.data_ptr())   # back-to-back, no host sync in between
            if not torch.equal(yd[:m], want):
                bad += 1
        print("%s env=%s tasks=%d split_rows=%d launches=%d -> %d bad" % (np.dtype(dtype).name, env, info["num_tasks"], info["num_split_rows"], launches, bad), flush=True)
        bad_total += bad
        p.close()
print("TOTAL BAD", bad_total)
sys.exit(1 if bad_total else 0)
