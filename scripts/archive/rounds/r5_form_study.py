"""Round 5: split (csr_split = 1) against pooled (csr_split = 2) units over the population — times and stream bytes side by side, to calibrate the byte-model rule that chooses between them."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, population_sweep as PS
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None
for key, wl, klass in PS.POPULATION:
    if only and key not in only: continue
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    v, x = G.compat_values(len(ci)), G.compat_x(n)
    tm = api.Tile_create(rows, n, nnz, rp, ci, v)
    balg = api.algorithmic_bytes(nnz, rows, n, 8)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
    res = {}
    for form in (1, 2):
        p = api.Plan(tm, rows, n, nnz, csr_split=form, placement_tries=1)
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=5, reps=30) for _ in range(3))
        i = p.info(); res[form] = (ms, i["stream_bytes"], i["entry_mode"], i["desc_bytes"])
        p.close()
    d, ia = api.plan_layout_digest(tm, rows, n, nnz)
    print("%-16s nnz/row %5.1f  split %.4f ms (%.3f B_alg, desc %2d, em %d)  pooled %.4f ms (%.3f B_alg, em %d)  time ratio %.3f  byte ratio %.3f  auto -> %d" % (
        key, nnz / rows, res[1][0], res[1][1] / balg, res[1][3], res[1][2], res[2][0], res[2][1] / balg, res[2][2], res[2][0] / res[1][0], res[2][1] / res[1][1], ia["csr_form"]), flush=True)
    api.Tile_destroy(tm); del xd, yd; torch.cuda.empty_cache()
