"""Times of one workload under a list of knob sets: python scripts/knob_time.py <workload> <f64|f32> "k=v,k=v" "k=v" ...   ("" = defaults); two repetitions, deterministic plans"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
wl = sys.argv[1]; dt = np.float32 if sys.argv[2] == "f32" else np.float64
m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
for rep in (1, 2):
    for spec in sys.argv[3:]:
        kw = {a.split("=")[0]: int(a.split("=")[1]) for a in spec.split(",") if a}
        p = api.Plan(tm, rows, n, nnz, **dict(dict(deterministic=1), **kw))
        ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=20, reps=100) for _ in range(3))
        i = p.info()
        print("%-16s rep %d %-40s %.4f ms  tasks %d stream_bytes %d" % (wl, rep, spec or "(defaults)", ms, i["num_tasks"], i["stream_bytes"]), flush=True)
        p.close()
