#!/usr/bin/env python3
"""Round 5: measure a POPULATION, not eleven matrices — the repo's answer to the authors' sweep (reference src/external/CSR5_cuda/bench0.sh:1-14 runs their
binary over a directory of SuiteSparse files; there is no network here, so the structures are generated).

26 structures, each >= 10 M nonzeros and > 256 MB of CSR-model bytes, in the class mix of the 173 matrices with >= 10 M nonzeros of the authors' list
(reference src/external/CSR5_cuda/2757-matrix.csv): structural FEM with 3 / 6 unknowns per node and shells (the largest group: audikw_1 :1252, ldoor :1268, bone010 :1453,
af_shell10 :1586, Serena ... Hook_1498 :2541-2546, Flan_1565 :2544), 2-D / 3-D meshes in natural and shuffled order (delaunay_n2x :2476-2479, hugebubbles :2480-2482),
road networks (:2509-2514), circuits (circuit5M :2276, Freescale1 :2277), KKT (nlpkkt :1901-1905), stencils, web-graph-like and R-MAT / Kronecker graphs (kron_g500 :2490-2494,
soc-LiveJournal1 :2285), plus band, band + random fill and one uniform random matrix as the worst case.

Per structure: DEFAULT plan, whole y against the scipy CSR product (exact: the reference driver's integer data), ms per SpMV (hip events, 50 launches), `frac` = SURVEY S8(d)'s
CSR-model bytes / time / 8 TB/s, `frac_min_bytes` = (values + x + y) / time / 8 TB/s, plan bytes / irreducible bytes, tile-format histogram, what the plan chose, seconds of
Tile_create and of plan creation (and how much of that was spent timing candidates).

    python scripts/population_sweep.py [--out profiles/r05_population.json] [--only name,name] [--dtype f64]
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (key, bench.py workload name, class)
POPULATION = [
    ("fem3_68", "fem3_68", "FEM 3 dof, natural order"), ("fem3s64_68", "fem3s64_68", "FEM 3 dof, shuffled in windows of 64"),
    ("fem6_46", "fem6_46", "FEM 6 dof, natural order"), ("fem6s64_46", "fem6s64_46", "FEM 6 dof, shuffled in windows of 64"),
    ("fem3_86", "fem3_86", "FEM 3 dof, natural order (Flan_1565 / Hook_1498 size)"), ("fem12_20", "fem12_20", "FEM, 12 unknowns per node (nd24k-like block rows)"),
    ("shell4_780", "shell4_780", "shell: 9-point quad mesh x 4 dof (af_shell-like)"), ("shell6s64_560", "shell6s64_560", "shell x 6 dof, shuffled in windows of 64"),
    ("tri2200", "tri2200", "2-D triangulation, natural order"), ("tri2200s4096", "tri2200s4096", "2-D triangulation, shuffled in windows of 4096 (delaunay-like)"),
    ("tet150", "tet150", "3-D tetrahedral mesh, natural order"), ("tet150s512", "tet150s512", "3-D tetrahedral mesh, shuffled in windows of 512"),
    ("road3400", "road3400", "road-network-like, natural order"), ("road3400s4096", "road3400s4096", "road-network-like, shuffled in windows of 4096"),
    ("circuit4m", "circuit4000000", "circuit-like"), ("nlpkkt160_f64", "nlpkkt160", "KKT (nlpkkt160 stand-in)"),
    ("lap3d256", "lap3d256", "7-point stencil 256^3"), ("laplacian4096", "laplacian4096", "5-point stencil 4096^2 (config 4)"), ("stencil27_160", "fem1_160", "27-point stencil 160^3"),
    ("band40_2m", "band40_2000000", "full band hbw 40 (dense tiles)"), ("bandrand4x3_2m", "bandrand4x3_2000000", "band + random fill"),
    ("powerlaw8m", "powerlaw8000000", "web-graph-like: host-local band + power-law"), ("plaw18_6m", "plaw18_6000000", "web-graph-like, heavier tail (exponent 1.8)"),
    ("rmat22x8", "rmat22x8", "R-MAT / Kronecker scale 22"), ("rmat21x16", "rmat21x16", "R-MAT / Kronecker scale 21, 16 edges per vertex"),
    ("uniform8_4m", "uniform8_4000000", "uniform random (worst case, not a SuiteSparse class)"),
]
LIVE_SUBSET = ["rmat22x8", "tri2200s4096", "tet150", "road3400", "shell4_780", "circuit4m"]   # what bench.py re-measures live in every run (fem3_68 / fem6_46 / fem3s64_68 are in its other_workloads)


def measure(key, wl, klass, dtype, torch, api, G, build_matrix, reps=50):
    import scipy.sparse as sp
    from tilespmv_amd.tile_matrix import field_array
    t0 = time.time()
    m, n, rp, ci, src = build_matrix(wl)
    t_gen = time.time() - t0
    rows = (m // 16) * 16; nnz = int(rp[rows])
    isz = np.dtype(dtype).itemsize
    v, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
    t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dtype); t_tc = time.time() - t0
    hist = np.bincount(field_array(tm, "Format", tm.tilenum), minlength=7).tolist()
    tiles = int(tm.tilenum)
    want = sp.csr_matrix((v[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64)
    t0 = time.time(); p = api.Plan(tm, rows, n, nnz); t_pc = time.time() - t0
    xd = torch.from_numpy(x).cuda(); yd = torch.full((rows + 16,), -1.0, dtype=torch.float64 if isz == 8 else torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
    ok = bool(np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), want))
    ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=10, reps=reps) for _ in range(2))
    i = p.info()
    b_alg = api.algorithmic_bytes(nnz, rows, n, isz); b_min = isz * (nnz + n + rows)
    rec = {"workload": wl, "class": klass, "source": src, "dtype": "f64" if isz == 8 else "f32", "rows": rows, "cols": n, "nnz": nnz, "nnz_per_row": round(nnz / rows, 2), "tiles": tiles,
           "tile_format_histogram[csr,coo,ell,hyb,dns,dnsrow,dnscol]": hist, "algorithmic_bytes": int(b_alg), "min_bytes": int(b_min), "plan_stream_bytes": i["stream_bytes"],
           "plan_over_irreducible_bytes": round(i["stream_bytes"] / b_min, 3), "plan_over_b_alg": round(i["stream_bytes"] / b_alg, 3),
           "ms_per_spmv": round(ms, 5), "gflops": round(2.0 * nnz / ms * 1e-6, 1), "frac": round(b_alg / ms * 1e-6 / 8000.0, 4), "frac_min_bytes": round(b_min / ms * 1e-6 / 8000.0, 4),
           "frac_by_plan_bytes": round(i["stream_bytes"] / ms * 1e-6 / 8000.0, 4), "check_whole_y_exact": "pass" if ok else "FAIL",
           "plan": {k: i[k] for k in ("csr_form", "entry_mode", "entry_ordered", "strip_cost", "num_tasks", "num_split_rows", "desc_bytes", "nt_stream", "brick_order", "x_panels", "x_panel_merge",
                                      "x_slice_passes", "placement_tries", "dense_mode", "scattered_entries")},
           "generate_seconds": round(t_gen, 2), "tile_create_seconds": round(t_tc, 3), "plan_create_seconds": round(t_pc, 3), "timed_choices_ms": round(i["timed_choices_us"] * 1e-3, 1)}
    p.close(); api.Tile_destroy(tm)
    # the same default plan prepared on the device (tilespmv_plan_create_from_csr: only the CSR arrays cross the bus), whole y checked the same way
    t0 = time.time(); pd = api.Plan.from_csr(rows, n, nnz, rp, ci, v, dtype=dtype); t_dev = time.time() - t0
    yd.fill_(-1.0)
    pd.spmv(xd.data_ptr(), yd.data_ptr(), st); torch.cuda.synchronize()
    idv = pd.info()
    rec["prepared_on_device"] = {"csr_to_plan_seconds": round(t_dev, 3), "tile_create_incl_csr_upload_seconds": round(idv["tile_create_us"] * 1e-6, 3), "timed_choices_ms": round(idv["timed_choices_us"] * 1e-3, 1),
                                 "same_form": bool(idv["csr_form"] == i["csr_form"] and idv["entry_mode"] == i["entry_mode"] and idv["num_tasks"] == i["num_tasks"]),
                                 "check_whole_y_exact": "pass" if bool(np.array_equal(yd.cpu().numpy()[:rows].astype(np.float64), want)) else "FAIL"}
    pd.close()
    del xd, yd
    return rec


def summarise(recs):
    ok = [r for r in recs.values() if "frac" in r]
    fr = sorted(r["frac"] for r in ok); fm = sorted(r["frac_min_bytes"] for r in ok)
    med = lambda a: (a[len(a) // 2] if len(a) % 2 else 0.5 * (a[len(a) // 2 - 1] + a[len(a) // 2])) if a else None
    return {"count": len(ok), "failed_checks": [k for k, r in recs.items() if r.get("check_whole_y_exact") != "pass"],
            "frac_median": med(fr), "frac_min": fr[0] if fr else None, "frac_max": fr[-1] if fr else None,
            "share_frac_ge_0.70": round(sum(f >= 0.70 for f in fr) / max(1, len(fr)), 3), "frac_min_bytes_median": med(fm),
            "share_frac_min_bytes_ge_0.60": round(sum(f >= 0.60 for f in fm) / max(1, len(fm)), 3),
            "plan_create_seconds_total": round(sum(r["plan_create_seconds"] for r in ok), 2), "timed_choices_ms_total": round(sum(r["timed_choices_ms"] for r in ok), 1),
            "host_preparation_seconds_total": round(sum(r["plan_create_seconds"] + r["tile_create_seconds"] for r in ok), 2),
            "device_preparation_seconds_total": round(sum(r["prepared_on_device"]["csr_to_plan_seconds"] for r in ok if "prepared_on_device" in r), 2),
            "device_prepared_checks_failed": [k for k, r in recs.items() if r.get("prepared_on_device", {}).get("check_whole_y_exact", "pass") != "pass"],
            "below_0.70": {k: r["frac"] for k, r in recs.items() if r.get("frac", 1) < 0.70}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_population.json"))
    ap.add_argument("--only", default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    a = ap.parse_args()
    import torch
    import bench
    from tilespmv_amd import api, generators as G
    assert torch.cuda.is_available()
    dtype = np.dtype(np.float64 if a.dtype == "f64" else np.float32)
    keep = set(a.only.split(",")) if a.only else None
    recs = {}
    t_all = time.time()
    for key, wl, klass in POPULATION:
        if keep and key not in keep:
            continue
        t0 = time.time()
        try:
            recs[key] = measure(key, wl, klass, dtype, torch, api, G, bench.build_matrix)
            r = recs[key]
            print("%-16s %-62s nnz %6.1f M  %.4f ms  frac %.3f  min %.3f  plan/irr %.2f  csr_form %d entry_mode %d  create %.2f s (timed %.0f ms)  %s  [%.0f s]" % (
                key, klass[:62], r["nnz"] / 1e6, r["ms_per_spmv"], r["frac"], r["frac_min_bytes"], r["plan_over_irreducible_bytes"], r["plan"]["csr_form"], r["plan"]["entry_mode"],
                r["plan_create_seconds"], r["timed_choices_ms"], r["check_whole_y_exact"], time.time() - t0), flush=True)
        except Exception as e:
            recs[key] = {"workload": wl, "class": klass, "error": repr(e)}
            print("%-16s ERROR %r" % (key, e), flush=True)
        torch.cuda.empty_cache()
    out = {"what": "population sweep, default plans, MI355X, %s, integer-valued data, whole y exact; frac = B_alg / t / 8 TB/s (SURVEY S8d)" % a.dtype,
           "measured": time.strftime("%Y-%m-%d %H:%M:%S"), "seconds": round(time.time() - t_all, 1), "summary": summarise(recs), "matrices": recs}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out["summary"]))


if __name__ == "__main__":
    main()
