"""Round 4: how much of the irregular class's time is the x gather's cache level?  Same rows, same entries per row, same plan
shape; only the number of COLUMNS (= the size of x the gathers spread over) changes: x that fits an XCD's 4-MB L2 (<= 256 k
columns in fp64) ... x of 64 MB.  B_alg barely moves (x is a few per cent of it), so the time difference is the gathers.
python scripts/archive/rounds/r4_gather_locality.py [rows] [per_row]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tilespmv_amd import api, generators as G

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
per_row = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print("uniform random, %d rows, %d per row, fp64; columns vary" % (rows, per_row), flush=True)
for cols in (131072, 262144, 524288, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
    m, n, rp, ci = G.uniform_per_row(rows, cols, per_row, 1)
    nnz = int(rp[rows])
    vals, x = G.compat_values(len(ci), np.float64), G.compat_x(n, np.float64)
    tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=np.float64)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
    balg = api.algorithmic_bytes(nnz, rows, n, 8)
    for kw in ({}, {"entry_ordered": 0}):
        p = api.Plan(tm, rows, n, nnz, **kw)
        t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(3))
        i = p.info()
        print("cols %8d (x = %5.1f MB)  nnz %9d  %-22s %.4f ms  B_alg/t %.2f TB/s = %.3f   plan bytes/t %.2f TB/s  mode %d strip %d tasks %d nt %d" % (
            cols, cols * 8e-6, nnz, str(kw), t, balg / t * 1e-9, balg / t * 1e-9 / 8, i["stream_bytes"] / t * 1e-9, i["entry_mode"], i["strip_cost"], i["num_tasks"], i["nt_stream"]), flush=True)
        p.close()
    api.Tile_destroy(tm)
    del xd, yd
