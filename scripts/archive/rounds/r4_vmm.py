"""Round 4: does the physical chunking of a large plan's blocks decide its placement state?  Plans of one Tile_matrix, built one after the other in ONE process
(three alive at a time), placement retry off; the blocks either plain hipMalloc or one virtual range mapped onto separately created physical chunks of 64 / 256 / 1024 MB
(experiment knob TILESPMV_ARENA_VMM_MB).  python scripts/archive/rounds/r4_vmm.py [workload] [f32] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "nlpkkt160"
dt = np.float32 if "f32" in sys.argv[2:] else np.float64
N = int([a for a in sys.argv[2:] if a.isdigit()][0]) if [a for a in sys.argv[2:] if a.isdigit()] else 6
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, src = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
want = None
print("%s (%s) %s, %d rows, %d nnz" % (wl, src, np.dtype(dt).name, rows, nnz), flush=True)
for rnd in range(2):
    for chunk in [int(c) for c in os.environ.get("VMM_CHUNKS", "0,256,64,1024,32,2").split(",")]:
        if chunk: os.environ["TILESPMV_ARENA_VMM_MB"] = str(chunk)
        else: os.environ.pop("TILESPMV_ARENA_VMM_MB", None)
        plans, out = [], []
        for i in range(N):
            p = api.Plan(tm, rows, n, nnz, placement_tries=1)
            t = min(p.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=20) for _ in range(3))
            if want is None: want = yd.clone()
            ok = bool(torch.equal(yd, want))
            out.append((t, ok))
            plans.append(p)
            if len(plans) > 3: plans.pop(0).close()
        for p in plans: p.close()
        ts = np.array([t for t, _ in out])
        print("round %d  %-28s ms: %s   %s" % (rnd, "hipMalloc blocks" if not chunk else "VMM chunks of %d MB" % chunk, " ".join("%.4f" % t for t in ts), "y equal" if all(o for _, o in out) else "Y DIFFERS"), flush=True)
