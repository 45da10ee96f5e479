"""Run the measured plan selection (tilespmv_plan_options.autotune / TILESPMV_AUTOTUNE) on the BASELINE workloads and a few
more, and keep what it saw: candidates, times, choice — profiles/autotune_rNN.json.  python scripts/autotune_receipts.py out.json"""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

out_path = sys.argv[1] if len(sys.argv) > 1 else "autotune.json"
sys.argv = sys.argv[:1]
import bench
work = [("laplacian4096", "f64"), ("scircuit", "f64"), ("webbase", "f64"), ("nlpkkt160", "f32"), ("nlpkkt160", "f64"), ("lap3d256", "f64"),
        ("band40_2000000", "f64"), ("powerlaw8000000", "f64"), ("powerlaw2000000", "f64"),
        # round 5: the mesh classes and the structures whose sweeps showed headroom no rule separates
        ("fem3_68", "f64"), ("fem6_46", "f64"), ("fem3s64_68", "f64"), ("fem1_160", "f64"), ("shell4_780", "f64"), ("tet150", "f64"), ("tet150s512", "f64"), ("circuit4000000", "f64")]
if os.environ.get("AUTOTUNE_ONLY"):
    work = [w for w in work if w[0] in os.environ["AUTOTUNE_ONLY"].split(",")]
res = []
for wl, dt in work:
    dtype = np.float32 if dt == "f32" else np.float64
    m, n, rp, ci, src = bench.build_matrix(wl)
    rows = (m // 16) * 16; nnz = int(rp[rows])
    vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
    tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype, hyb=(wl == "scircuit"))
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
    log = tempfile.mktemp(suffix=".jsonl")
    os.environ["TILESPMV_AUTOTUNE_LOG"] = log
    p_auto = api.Plan(tm, rows, n, nnz, autotune=True)
    os.environ.pop("TILESPMV_AUTOTUNE_LOG")
    p_def = api.Plan(tm, rows, n, nnz)
    t_def, t_auto = [], []
    for _ in range(5):
        t_def.append(p_def.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=30)); t_auto.append(p_auto.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=30))
    rec = json.loads(open(log).read().strip().splitlines()[-1])
    rec.update({"workload": wl, "dtype": dt, "source": src, "default_plan_ms": round(min(t_def), 5), "autotuned_plan_ms": round(min(t_auto), 5),
                "gain": round(min(t_def) / min(t_auto), 4)})
    res.append(rec)
    rec["autotuned_plan_facts"] = {k: p_auto.info()[k] for k in ("csr_form", "entry_mode", "entry_ordered", "strip_cost", "wg_strips", "brick_order", "dense_mode", "num_tasks")}
    print(wl, dt, "default %.5f ms  autotuned %.5f ms  (x %.3f)  choice %s  facts %s" % (min(t_def), min(t_auto), min(t_def) / min(t_auto), rec["choice"], rec["autotuned_plan_facts"]), flush=True)
    p_auto.close(); p_def.close(); api.Tile_destroy(tm)
    del xd, yd
json.dump({"what": "tilespmv_plan_create with measured selection on every workload: candidates timed at plan creation, choice, and the default vs tuned plan re-timed afterwards (min of 5 x 30)", "results": res}, open(out_path, "w"), indent=1)
