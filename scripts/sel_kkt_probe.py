import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tilespmv_amd import api, generators as G
from tilespmv_amd.tile_matrix import field_array
sys.argv = sys.argv[:1]
import bench
for dtype in (np.float64, np.float32):
    m, n, rp, ci, src = bench.build_matrix("nlpkkt160")
    rows = (m // 16) * 16; nnz = int(rp[rows])
    vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    tms = {"ref": api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype), "cdna4": api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype, cdna4=True)}
    fa, fb = field_array(tms["ref"], "Format", tms["ref"].tilenum), field_array(tms["cdna4"], "Format", tms["cdna4"].tilenum)
    wa, wb = field_array(tms["ref"], "tilewidth", tms["ref"].tilenum), field_array(tms["cdna4"], "tilewidth", tms["cdna4"].tilenum)
    print("tiles with another format:", int(np.count_nonzero(fa != fb)), "another width:", int(np.count_nonzero(wa != wb)), flush=True)
    plans = [(k + str(i), api.Plan(tms[k], rows, n, nnz)) for i in range(2) for k in ("ref", "cdna4")]
    res = {k: [] for k, _ in plans}
    for rnd in range(5):
        for k, p in plans:
            res[k].append(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20))
    for k, p in plans:
        i = p.info(); print(np.dtype(dtype).name, k, "min %.4f ms" % min(res[k]), "tasks", i["num_tasks"], "stream", i["stream_bytes"], "device_bytes", i["device_bytes"], flush=True)
    for k, p in plans: p.close()
    del xd, yd
