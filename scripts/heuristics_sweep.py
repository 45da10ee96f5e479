"""Do the plan defaults (entry mode, strip size, COO / dense mode) hold up on matrices they were not tuned on?
For a spread of synthetic structures: default plan vs the measured selection (autotune), and the time of every candidate.
python scripts/heuristics_sweep.py out.json"""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tilespmv_amd import api, generators as G

out_path = sys.argv[1] if len(sys.argv) > 1 else "heuristics.json"


def rmat(scale, ef, seed, a=0.57, b=0.19, c=0.19):
    rng = np.random.default_rng(seed)
    n = 1 << scale; m = ef * n
    ri = np.zeros(m, np.int64); ci = np.zeros(m, np.int64)
    for lvl in range(scale):
        u = rng.random(m)
        rbit = (u >= a + b).astype(np.int64); cbit = ((u >= a) & (u < a + b) | (u >= a + b + c)).astype(np.int64)
        ri = (ri << 1) | rbit; ci = (ci << 1) | cbit
    return G.from_coo(n, n, ri, ci)


def stencil9(n):
    N = n * n; idx = np.arange(N, dtype=np.int64); i, j = idx // n, idx % n
    cand, mask = [], []
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            cand.append(idx + di * n + dj); mask.append((i + di >= 0) & (i + di < n) & (j + dj >= 0) & (j + dj < n))
    rp, ci = G._from_mask(np.stack(cand, 1), np.stack(mask, 1))
    return N, N, rp, ci


def band_plus_random(n, hbw, extra, seed):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n, dtype=np.int64), 2 * hbw + 1); c = r + np.tile(np.arange(-hbw, hbw + 1), n)
    ok = (c >= 0) & (c < n)
    rr = rng.integers(0, n, extra * n); cc = rng.integers(0, n, extra * n)
    return G.from_coo(n, n, np.concatenate([r[ok], rr]), np.concatenate([c[ok], cc]))


def block_diag_plus_sparse(nb, bs, extra, seed):
    rng = np.random.default_rng(seed)
    n = nb * bs
    lr, lc = np.meshgrid(np.arange(bs), np.arange(bs), indexing="ij")
    keep = rng.random((nb, bs, bs)) < 0.6
    b, i, j = np.nonzero(keep)
    rr = rng.integers(0, n, extra * n); cc = rng.integers(0, n, extra * n)
    return G.from_coo(n, n, np.concatenate([b * bs + i, rr]), np.concatenate([b * bs + j, cc]))


work = [
    ("uniform random 1M x 1M, 8/row", lambda: G.random_uniform(1 << 20, 1 << 20, 8.0 / (1 << 20), 1)),
    ("uniform random 250k, 40/row", lambda: G.random_uniform(250000, 250000, 40.0 / 250000, 2)),
    ("R-MAT scale 20, 8 edges/vertex", lambda: rmat(20, 8, 3)),
    ("R-MAT scale 17, 16 edges/vertex", lambda: rmat(17, 16, 4)),
    ("9-point stencil 2048^2", lambda: stencil9(2048)),
    ("band hbw 4 + 3 random/row, 2M", lambda: band_plus_random(2000000, 4, 3, 5)),
    ("band hbw 12, 1M", lambda: G.band(1000000, 12)),
    ("block-diagonal 24x24 (60 %) + 2 random/row, 600k", lambda: block_diag_plus_sparse(25000, 24, 2, 6)),
    ("circuit-like 1M", lambda: G.circuit_like(1000000, seed=7)),
    ("circuit-like 40k", lambda: G.circuit_like(40000, seed=8)),
    ("power-law 300k", lambda: G.powerlaw(300000, seed=9)),
    ("power-law 4M", lambda: G.powerlaw(4000000, seed=10)),
    ("KKT-like 64^3 (round-1 generator)", lambda: G.kkt_like(64)),
    ("7-point 128^3", lambda: G.laplacian7pt(128)),
]
res = []
for name, gen in work:
    m, n, rp, ci = gen()
    rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in (np.float64,):
        vals, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda")
        log = tempfile.mktemp(suffix=".jsonl"); os.environ["TILESPMV_AUTOTUNE_LOG"] = log
        p_auto = api.Plan(tm, rows, n, nnz, autotune=True)
        os.environ.pop("TILESPMV_AUTOTUNE_LOG")
        p_def = api.Plan(tm, rows, n, nnz)
        td, ta = [], []
        for _ in range(5):
            td.append(p_def.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=30)); ta.append(p_auto.time(xd.data_ptr(), yd.data_ptr(), warmup=3, reps=30))
        rec = json.loads(open(log).read().strip().splitlines()[-1])
        i = p_def.info()
        balg = api.algorithmic_bytes(nnz, rows, n, 8)
        rec.update({"matrix": name, "rows": rows, "nnz": nnz, "default_plan_ms": round(min(td), 5), "autotuned_plan_ms": round(min(ta), 5), "gain": round(min(td) / min(ta), 4),
                    "default": {"entry_mode": i["entry_mode"], "strip_cost": i["strip_cost"], "coo_mode": i["coo_mode"], "dense_mode": i["dense_mode"], "ordered": i["entry_ordered"]},
                    "default_frac_of_8TBps": round(balg / min(td) * 1e-6 / 8000, 4)})
        res.append(rec)
        print("%-50s nnz %9d  default %.5f ms (%.0f %% of 8 TB/s, mode %d, strip %d)  tuned %.5f  gain %.3f  choice %s" % (
            name, nnz, min(td), 100 * rec["default_frac_of_8TBps"], i["entry_mode"], i["strip_cost"], min(ta), rec["gain"], {k: rec["choice"][k] for k in ("entry_mode", "ordered", "strip_cost", "coo_mode", "dense_mode")}), flush=True)
        p_auto.close(); p_def.close(); api.Tile_destroy(tm)
        del xd, yd
json.dump({"what": "default plan vs measured selection on matrices the defaults were not tuned on", "results": res}, open(out_path, "w"), indent=1)
