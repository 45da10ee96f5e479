"""Round 5 (second session): what the measured selection (autotune = 1) costs on the two preparation paths — host: every candidate re-laid-out from the host Tile_matrix and uploaded;
device: every candidate built by kernels from the one device-resident tiled matrix."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v = G.compat_values(len(ci)); x = G.compat_x(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    for rep in range(2):
        t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, v); ph = api.Plan(tm, rows, n, nnz, autotune=True); t_h = time.time() - t0
        t0 = time.time(); pd = api.Plan.from_csr(rows, n, nnz, rp, ci, v, autotune=True); t_d = time.time() - t0
        ms_h = min(ph.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(2)); ms_d = min(pd.time(xd.data_ptr(), yd.data_ptr(), 0, 10, 40) for _ in range(2))
        ih, idv = ph.info(), pd.info()
        print("%-14s rep %d: autotuned plan from the host path %.2f s (%.4f ms / SpMV: form %d mode %d strip %d) | from the device path %.2f s (%.4f ms / SpMV: form %d mode %d strip %d)" % (
              wl, rep, t_h, ms_h, ih["csr_form"], ih["entry_mode"], ih["strip_cost"], t_d, ms_d, idv["csr_form"], idv["entry_mode"], idv["strip_cost"]), flush=True)
        ph.close(); pd.close(); api.Tile_destroy(tm)
