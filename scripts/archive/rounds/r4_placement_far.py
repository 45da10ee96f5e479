"""Placement retry policies on the KKT stand-in (fp64): N plans per setting in ONE process (three alive at a time), up to T placements each; what the retry kept, and after how many tries.
python scripts/archive/rounds/r4_placement_far.py   (FAR_SETTINGS="A=1,B=2;C=3" FAR_PLANS=6 FAR_TRIES=12)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tilespmv_amd import api, generators as G
sys.argv = sys.argv[:1]
import bench
wl = os.environ.get("FAR_WORKLOAD", "nlpkkt160")
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), np.float64), G.compat_x(n, np.float64)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=np.float64)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
N, T = int(os.environ.get("FAR_PLANS", "6")), int(os.environ.get("FAR_TRIES", "12"))
print(wl, "plans per setting", N, "placements per plan up to", T, flush=True)
for setting in os.environ.get("FAR_SETTINGS", "TILESPMV_RETRY_SPACER_MB=0;TILESPMV_RETRY_SPACER_MB=16;TILESPMV_RETRY_SPACER_MB=100;TILESPMV_RETRY_SPACER_MB=256").split(";"):
    kv = dict(s.split("=") for s in setting.split(","))
    os.environ.update(kv)
    plans, out = [], []
    for i in range(N):
        p = api.Plan(tm, rows, n, nnz, placement_tries=T)
        t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(3))
        out.append((t, p.info()["placement_tries"], p.info()["build_us"] * 1e-6))
        plans.append(p)
        if len(plans) > 3: plans.pop(0).close()
    for p in plans: p.close()
    for k in kv: os.environ.pop(k)
    print("%-44s ms: %s | tries: %s | build s: %s" % (setting, " ".join("%.4f" % t for t, _, _ in out), " ".join(str(k) for _, k, _ in out), " ".join("%.1f" % b for _, _, b in out)), flush=True)
