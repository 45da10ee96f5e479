"""Round 5: pooled units (csr_split = 2) — whole-y exact check against scipy on small matrices in every entry mode, then timing of the FEM class split vs pooled.
python scripts/archive/rounds/r5_pool_check.py [check|time|all]"""
import os, sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tilespmv_amd import api, generators as G

what = sys.argv[1] if len(sys.argv) > 1 else "all"
st = torch.cuda.current_stream().cuda_stream


def run(tm, rows, n, nnz, x, dt, **kw):
    p = api.Plan(tm, rows, n, nnz, **kw)
    xd = torch.from_numpy(x).cuda(); yd = torch.full((rows + 16,), -7.0, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
    p.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
    return p, xd, yd


if what in ("check", "all"):
    mats = {"fem3_12": G.fem_hex(12, 12, 12, 3), "fem3s_14": G.fem_hex(14, 11, 9, 3, shuffle=16), "fem6_9": G.fem_hex(9, 9, 9, 6), "fem2_odd": G.fem_hex(13, 7, 5, 2),
            "allfmt": G.all_formats(12, 7), "allfmt_pad": G.all_formats(9, 3, cols_pad=5), "kkt12": G.kkt_like(12), "lap64": G.laplacian5pt(64), "powerlaw": G.powerlaw(60000, seed=2),
            "band40": G.band(3000, 40), "bandrand": G.band_plus_random(40000, 4, 3, 5), "rmat14": G.rmat(14, 8, 3), "blockdiag": G.block_diag_plus_sparse(300, 24, 2, 6)}
    bad = 0; n_plans = 0
    for name, (m, n, rp, ci) in mats.items():
        rows = (m // 16) * 16; nnz = int(rp[rows])
        for dt in (np.float64, np.float32):
            v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
            tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt, hyb=name.startswith("allfmt"))
            want = sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64)
            knobsets = [dict(), dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2), dict(entry_mode=2, entry_ordered=1), dict(entry_mode=2, entry_ordered=0),
                        dict(strip_cost=64, split_above=200), dict(entry_mode=2, strip_cost=100, split_above=300, split_cap=300), dict(entry_mode=0, fix_inline=0, split_above=150, strip_cost=50),
                        dict(dense_mode=api.DENSE_MFMA), dict(dense_mode=api.DENSE_VALU), dict(coo_mode=api.COO_FALLBACK), dict(xcd_remap=0), dict(nt_stream=1), dict(entry_mode=2, nt_stream=1),
                        dict(entry_mode=2, x_panel_kb=4, x_panel_merge=1), dict(entry_mode=2, x_panel_kb=4, x_slice_passes=1), dict(x_window=2), dict(lds_pad=8192),
                        dict(tilerow_begin=3, tilerow_end=max(4, (rows // 16) // 2))]
            for kw in knobsets:
                kw = dict(kw, csr_split=2)
                p, xd, yd = run(tm, rows, n, nnz, x, dt, **kw)
                info = p.info()
                got = yd.cpu().numpy().astype(np.float64)
                r0, r1 = 16 * kw.get("tilerow_begin", 0), (16 * kw["tilerow_end"] if "tilerow_end" in kw else rows)
                ok = np.array_equal(got[r0:r1], want[r0:r1]) and (r0 == 0 or np.all(got[:r0] == -7.0)) and np.all(got[r1:rows + 16] == -7.0) if (r0, r1) != (0, rows) else np.array_equal(got[:rows], want)
                n_plans += 1
                if not ok or info["csr_form"] != 2:
                    bad += 1
                    d = np.nonzero(got[r0:r1] != want[r0:r1])[0]
                    print("MISMATCH", name, dt.__name__, kw, "csr_form", info["csr_form"], "entry_mode", info["entry_mode"], "bad rows", len(d), d[:8] + r0)
                # SpMM goes one right-hand side at a time on pooled plans
                if kw.get("entry_mode") == 0 and "tilerow_begin" not in kw:
                    X = np.stack([x, 2 * x], axis=1).copy(); Xd = torch.from_numpy(X).cuda(); Yd = torch.zeros((rows + 16, 2), dtype=yd.dtype, device="cuda")
                    p.spmm(Xd.data_ptr(), Yd.data_ptr(), 2, st); torch.cuda.synchronize()
                    Y = Yd.cpu().numpy().astype(np.float64)
                    if not (np.array_equal(Y[:rows, 0], want) and np.array_equal(Y[:rows, 1], 2 * want)):
                        bad += 1; print("SPMM MISMATCH", name, dt.__name__, kw)
                p.close()
            api.Tile_destroy(tm)
        print("checked", name, flush=True)
    print("pooled check: %d plans, %d bad" % (n_plans, bad), flush=True)
    if bad:
        sys.exit(1)

if what in ("time", "all"):
    for wl, gen in (("fem3_68", lambda: G.fem_hex(68, 68, 68, 3)), ("fem6_46", lambda: G.fem_hex(46, 46, 46, 6)), ("fem3s64_68", lambda: G.fem_hex(68, 68, 68, 3, shuffle=64))):
        m, n, rp, ci = gen()
        rows = (m // 16) * 16; nnz = int(rp[rows])
        for dt in (np.float64, np.float32):
            v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
            tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
            want = sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64)
            balg = api.algorithmic_bytes(nnz, rows, n, np.dtype(dt).itemsize); bmin = np.dtype(dt).itemsize * (nnz + n + rows)
            for kw in (dict(csr_split=1), dict(csr_split=2), dict(csr_split=2, entry_mode=0), dict(csr_split=2, entry_mode=2, entry_ordered=0), dict(csr_split=2, entry_mode=0, strip_cost=800), dict(csr_split=2, entry_mode=0, strip_cost=2400), dict(csr_split=2, x_window=0), dict()):
                t0 = time.time()
                p, xd, yd = run(tm, rows, n, nnz, x, dt, **kw)
                tc = time.time() - t0
                ok = np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), want)
                ms = p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=10, reps=50)
                i = p.info()
                print("%-11s %s %-60s %.4f ms frac %.3f min %.3f plan/B_alg %.3f csr_form %d entry_mode %d tasks %d brick %d nt %d create %.2fs %s" % (
                    wl, dt.__name__[5:], kw, ms, balg / ms * 1e-6 / 8000, bmin / ms * 1e-6 / 8000, i["stream_bytes"] / balg, i["csr_form"], i["entry_mode"], i["num_tasks"], i["brick_order"], i["nt_stream"], tc, "ok" if ok else "WRONG"), flush=True)
                p.close()
            api.Tile_destroy(tm)
