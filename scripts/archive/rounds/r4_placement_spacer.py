"""Does WHERE a plan's blocks land decide its state?  One Tile_matrix, plans built one after the other (each destroyed before the next), with an unused allocation of S MB in front of
every arena block (TILESPMV_ARENA_SPACER_MB) and with different block sizes (TILESPMV_ARENA_MB): python scripts/archive/rounds/r4_placement_spacer.py [workload] [f32]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
dt = np.float32 if "f32" in sys.argv[2:] else np.float64
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
print("%s %s" % (wl, np.dtype(dt).name), flush=True)
def one(env):
    for k, v in env.items(): os.environ[k] = str(v)
    p = api.Plan(tm, rows, n, nnz, placement_tries=1)
    t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(3))
    p.close()
    for k in env: os.environ.pop(k)
    return t
import json
def show(label, env, n=2):
    print("%-64s %s" % (label, " ".join("%.4f" % one(env) for _ in range(n))), flush=True)
show("no spacer", {}, 3)
for gb in (4, 8, 16, 24, 32, 40):                      # (at most 5 blocks x 40 GB: well inside the 288 GB)
    show("%2d GB in front of EVERY block" % gb, {"TILESPMV_ARENA_SPACER_MB": gb * 1024})
for gb in (8, 32, 64, 128, 200):
    show("%3d GB in front of the FIRST block only (whole plan shifted)" % gb, {"TILESPMV_ARENA_SPACER_MB": gb * 1024, "TILESPMV_ARENA_SPACER_FIRST": 1})
show("no spacer again", {}, 3)
for mb in (64, 1024, 4096):
    show("arena blocks of %5d MB" % mb, {"TILESPMV_ARENA_MB": mb}, 3)
