"""How many nonzeros stay on entry lists, absorb off / on: python scripts/list_entries_probe.py wl,wl"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v, x = G.compat_values(len(ci), np.float64), G.compat_x(n, np.float64)
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=np.float64)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    hist = np.bincount(np.frombuffer(tm.Format_array(), dtype=np.uint8) if hasattr(tm, "Format_array") else np.zeros(1, np.uint8), minlength=7)
    out = []
    for a in (0, 1):
        p = api.Plan(tm, rows, n, nnz, deterministic=1, absorb=a)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3))
        i = p.info(); out.append((i["list_entries"], ms, i["csr_form"], i["entry_mode"], i["stream_bytes"]))
        p.close()
    print("%-16s nnz %10d  list entries %9d -> %9d (%.1f %% -> %.1f %% of nnz)  ms %.4f -> %.4f  csr_form %d entry_mode %d  stream %.1f -> %.1f MB" % (wl, nnz, out[0][0], out[1][0], 100.0 * out[0][0] / nnz, 100.0 * out[1][0] / nnz, out[0][1], out[1][1], out[1][2], out[1][3], out[0][4] / 1e6, out[1][4] / 1e6), flush=True)
    api.Tile_destroy(tm)
