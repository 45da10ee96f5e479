#!/usr/bin/env python3
"""bench.py — y = A*x throughput of the HIP TileSpMV engine on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--workload laplacian4096] [--dtype f64]

A "step" is one SpMV over the whole (row-partitioned) matrix.  Inputs are resident in HBM when
the timed region starts.  Rank 0 prints ONE JSON line.  Metric/config follow BASELINE.json:
fp64 SpMV GFLOP/s + achieved (algorithmic) HBM GB/s as a fraction of the 8 TB/s roofline.

Workloads (synthetic stand-ins unless $TILESPMV_MATRIX_DIR/<name>.mtx exists):
  laplacian4096  5-pt Laplacian on a 4096^2 grid, 16.7 M rows, 83.9 M nnz  (config 4; default —
                 the >= 10 M-nnz fp64 case the roofline target is quoted on, fits one GPU)
  scircuit       circuit-like, 171 k rows, ~1 M nnz (config 2)       webbase   power-law 1 M rows (config 3)
  nlpkkt160      KKT-like, 8.2 M rows, 1.66e8 nnz, fp32 by default (config 5)
Multi-GPU: contiguous nnz-balanced tile-row blocks, one rank per GPU, x replicated, y left
sharded (the SpMV needs no collective: SURVEY.md §8e) => "scaling": "strong" on the fixed matrix;
--combine allgather|allreduce adds the RCCL y combine to every step, and the default run also
reports both combine variants beside the headline value when N > 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_matrix(name):
    from tilespmv_amd import api, generators as G
    d = os.environ.get("TILESPMV_MATRIX_DIR")
    real = {"laplacian4096": None, "scircuit": "scircuit", "webbase": "webbase-1M", "nlpkkt160": "nlpkkt160"}.get(name)
    if d and real and os.path.exists(os.path.join(d, real + ".mtx")):
        r = api.mmio_allinone(os.path.join(d, real + ".mtx"))
        return r["m"], r["n"], r["rowptr"], r["colidx"], "file:" + real + ".mtx"
    if name == "laplacian4096":
        return G.laplacian5pt(4096) + ("synthetic 5-pt Laplacian 4096^2",)
    if name.startswith("lap3d"):
        return G.laplacian7pt(int(name[5:])) + ("synthetic 7-pt Laplacian on a cube",)
    if name.startswith("powerlaw"):
        return G.powerlaw(int(name[8:]), seed=2) + ("synthetic power-law",)
    if name.startswith("laplacian"):
        return G.laplacian5pt(int(name[len("laplacian"):])) + ("synthetic 5-pt Laplacian",)
    if name.startswith("band"):  # e.g. band40_2000000: full band, half-bandwidth 40 (dense-tile dominated)
        hbw, nn = name[4:].split("_")
        return G.band(int(nn), int(hbw)) + ("synthetic full band hbw=%s" % hbw,)
    if name == "scircuit":
        return G.circuit_like(170998, seed=1) + ("synthetic circuit-like stand-in for scircuit",)
    if name == "webbase":
        return G.powerlaw(1000005, seed=2) + ("synthetic power-law stand-in for webbase-1M",)
    if name == "nlpkkt160":
        return G.kkt_like(160, seed=5) + ("synthetic KKT-like stand-in for nlpkkt160",)
    raise SystemExit("unknown workload " + name)


def cpu_baseline(rows, cols, rowptr, colidx, vals, x, dtype, budget_s=12.0):
    """The CPU path of the reference timed on this box's host cores (rank 0, N=1 only), on a
    bounded sample of the same workload: its first <= 1,048,576 rows.  kind = "reference" when
    the prebuilt oracle/_ref (the reference's own headers) is present, else "port" (our C
    restatement).  Single-threaded like the reference (src/tilespmv_cpu.h:125).  Timed: the whole
    tilespmv_cpu call (schedule arrays + serial tile SpMV + self-check), Tile_create excluded."""
    from oracle.oracle import CpuImpl, available
    kind = "ref" if available("ref", dtype) else "oracle"
    impl = CpuImpl(kind, dtype)
    srows = min(rows, 1 << 20)
    nz = int(rowptr[srows])
    rp, ci, v = rowptr[:srows + 1], colidx[:nz], vals[:nz]
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); sys.stdout.flush(); os.dup2(devnull, 1)
    try:
        tm = impl.tile_create(srows, cols, nz, rp, ci, v)
        yg = impl.csr_spmv(srows, rp, ci, v, x)
        times = []
        t_end = time.time() + budget_s
        while len(times) < 3 or (time.time() < t_end and len(times) < 40):
            t0 = time.perf_counter()
            impl.spmv(tm, srows, cols, nz, rp, ci, v, x, yg)
            times.append(time.perf_counter() - t0)
    finally:
        sys.stdout.flush(); os.dup2(saved, 1); os.close(devnull); os.close(saved)
    # the same tile SpMV over all host cores (OpenMP over tile-rows; our restatement, bit-identical y): extra information
    allc = None
    try:
        O = CpuImpl("oracle", dtype)
        tmo = O.tile_create(srows, cols, nz, rp, ci, v)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); _, nthr = O.spmv_all_cores(tmo, srows, cols, x); ts.append(time.perf_counter() - t0)
        allc = {"value": round(2.0 * nz / float(np.median(ts[2:])) * 1e-9, 3), "unit": "GFLOP/s", "cores": int(nthr), "kind": "port (OpenMP over tile-rows)"}
    except Exception as e:
        allc = {"error": repr(e)}
    t = float(np.median(times))
    return {"all_host_cores": allc, "value": round(2.0 * nz / t * 1e-9, 4), "unit": "GFLOP/s", "cores": 1,
            "kind": "reference" if kind == "ref" else "port", "seconds_per_spmv": round(t, 6), "runs": len(times),
            "sample": "first %d rows (%d nnz) of the workload matrix; tilespmv_cpu whole call, median of %d runs" % (srows, nz, len(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="laplacian4096")
    ap.add_argument("--dtype", default=None, choices=[None, "f64", "f32"])
    ap.add_argument("--combine", default="none", choices=["none", "allgather", "allreduce"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo only to rehearse the N>1 path on a single GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tilespmv_amd import api, generators as G
    from tilespmv_amd.dist import ShardedSpMV

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:  # one process per GPU shares the host cores: keep the preprocessing threads per rank modest
        os.environ.setdefault("TILESPMV_NUM_THREADS", str(max(2, min(16, (os.cpu_count() or 16) // world))))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    dtype = np.dtype(np.float32 if (args.dtype == "f32" or (args.dtype is None and args.workload == "nlpkkt160")) else np.float64)
    tdtype = torch.float64 if dtype == np.float64 else torch.float32
    t0 = time.time()
    m, n, rowptr, colidx, source = build_matrix(args.workload)
    rows = (m // 16) * 16                      # the driver rule of the reference (src/main.cu:71)
    nnz = int(rowptr[rows])
    vals, x = G.compat_values(len(colidx), dtype), G.compat_x(n, dtype)   # reference's synthetic data (src/main.cu:68-69,:93-97)
    t_gen = time.time() - t0

    t0 = time.time()
    sh = ShardedSpMV(rank, world, rows, n, rowptr, colidx, vals, dtype)
    t_prep = time.time() - t0
    info = sh.local.info()
    stream = torch.cuda.current_stream()
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros(rows + 16, dtype=tdtype, device="cuda")

    def step(combine):
        sh.spmv(xd, yd, stream.cuda_stream)
        if combine != "none":
            sh.combine(yd, combine)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run(combine, count):  # exactly `count` steps, back to back on the launch stream
        if combine == "none":
            sh.spmv(xd, yd, stream.cuda_stream, count=count)   # one C call issuing `count` launches
        else:
            for _ in range(count):
                step(combine)

    def timed(combine, steps, warmup):
        run(combine, warmup)
        sync_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        run(combine, steps)
        e1.record(stream)
        sync_all()
        wall = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1)          # HIP events on the launch stream, over the timed region
        if world > 1:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        return wall, dev_ms

    # parity spot check of the resident plan before timing (exact: integer-valued data)
    check = "skipped"
    if not args.no_check:
        step("none"); torch.cuda.synchronize()
        rs = np.unique(np.concatenate([np.arange(sh.r0, min(sh.r0 + 4096, sh.r1)), np.random.default_rng(rank).integers(sh.r0, max(sh.r1, sh.r0 + 1), 20000)]))
        rs = rs[rs < sh.r1]
        got = yd[torch.from_numpy(rs).cuda()].cpu().numpy()
        want = np.array([np.dot(vals[rowptr[r]:rowptr[r + 1]].astype(np.float64), x[colidx[rowptr[r]:rowptr[r + 1]]].astype(np.float64)) for r in rs[:6000]])
        ok = np.array_equal(got[:len(want)].astype(np.float64), want)
        check = "pass" if ok else "FAIL"
        if not ok:
            raise SystemExit("bench.py: HIP result differs from the CSR golden on sampled rows")

    wall, dev_ms = timed(args.combine, args.steps, args.warmup)
    ms_per_step = wall * 1e3 / args.steps
    flops = 2.0 * nnz
    b_alg_total = api.algorithmic_bytes(nnz, rows, n, dtype.itemsize)
    value = flops / (wall / args.steps) * 1e-9

    # roofline of the dominant kernel (k_tiles_direct): algorithmic bytes of THIS rank's launch
    # divided by its average duration, from HIP events on the launch stream over the timed region.
    b_alg_launch = api.algorithmic_bytes(sh.local_nnz, sh.local_rows, n, dtype.itemsize)
    kernel_ms = dev_ms / args.steps
    achieved = b_alg_launch / (kernel_ms * 1e-3) * 1e-9
    traffic = None
    tj = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (args.workload, "f64" if dtype == np.float64 else "f32"))
    if world == 1 and os.path.exists(tj):
        traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "kernel": "k_units" if info["kernel"] == 2 else "k_tiles_direct", "kernel_ms": round(kernel_ms, 5), "algorithmic_bytes_per_launch": int(b_alg_launch),
                "plan_stream_bytes_per_launch": info["stream_bytes"], "timing": "hip events on the launch stream, timed region"}

    # measured device ceilings beside the 8 TB/s spec figure: read-only and copy streams of 1 GiB buffers
    if rank == 0 and world == 1:
        try:
            big = torch.ones(1 << 27, dtype=torch.float64, device="cuda"); dst = torch.empty_like(big)
            def _bw(f, nbytes):
                for _ in range(3): f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); [f() for _ in range(10)]; b.record(); torch.cuda.synchronize()
                return nbytes * 10 / (a.elapsed_time(b) * 1e-3) * 1e-9
            rd = _bw(lambda: torch.sum(big), big.numel() * 8); cp = _bw(lambda: dst.copy_(big), 2 * big.numel() * 8)
            roofline["measured_ceilings_gbps"] = {"read_only_stream": round(rd, 0), "copy_stream_read_plus_write": round(cp, 0)}
            if traffic:
                roofline["actual_traffic_gbps"] = round(traffic / (kernel_ms * 1e-3) * 1e-9, 1)
                roofline["actual_traffic_over_read_ceiling"] = round(traffic / (kernel_ms * 1e-3) * 1e-9 / rd, 4)
            del big, dst
        except Exception as e:
            roofline["measured_ceilings_gbps"] = {"error": repr(e)}

    extra = {}
    if world > 1 and args.combine == "none":
        for mode in ("allgather", "allreduce"):
            w2, _ = timed(mode, max(5, args.steps // 10), 3)
            extra[mode] = {"ms_per_step": round(w2 * 1e3 / max(5, args.steps // 10), 5), "gflops": round(flops / (w2 / max(5, args.steps // 10)) * 1e-9, 2)}
            if not args.no_check:  # after the combine every rank holds the whole y: sampled rows of ALL shards, exact
                ra = np.unique(np.random.default_rng(100 + rank).integers(0, rows, 3000))
                got = yd[torch.from_numpy(ra).cuda()].cpu().numpy().astype(np.float64)
                want = np.array([np.dot(vals[rowptr[r]:rowptr[r + 1]].astype(np.float64), x[colidx[rowptr[r]:rowptr[r + 1]]].astype(np.float64)) for r in ra])
                okc = torch.tensor([1.0 if np.array_equal(got, want) else 0.0], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(okc, op=dist.ReduceOp.MIN)
                extra[mode]["check_full_y_on_every_rank"] = "pass" if float(okc[0]) == 1.0 else "FAIL"

    out = {
        "metric": "fp%d SpMV GFLOP/s (y = A*x, tiled format)" % (dtype.itemsize * 8), "value": round(value, 2), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if dtype == np.float64 else "f32", "data": "synthetic (reference driver data: val[i]=i%10, x[i]=i%10)",
        "config": {"workload": args.workload, "source": source, "rows": rows, "cols": n, "nnz": nnz,
                   "partition": "tile-row blocks, nnz-balanced, %d rank(s)" % world, "y_combine": args.combine, "backend": args.backend if world > 1 else None,
                   "tiles": info["tiles"], "coo_mode": info["coo_mode"], "dense_mode": info["dense_mode"]},
        "hbm_gbps_algorithmic": round(b_alg_total / (wall / args.steps) * 1e-9, 1),
        "hbm_roofline_frac": round(b_alg_total / (wall / args.steps) * 1e-9 / (HBM_PEAK_GBPS * world), 4),
        "roofline": roofline, "check": check,
        "prep_seconds": {"generate": round(t_gen, 2), "tile_create_and_upload": round(t_prep, 2)},
    }
    if extra:
        out["with_y_combine"] = extra
    # the other BASELINE configs (stand-ins) at one GPU, reported beside the headline; never `value`:
    # configs 2-3 are cache-resident (launch/latency-bound), config 5 is the fp32 HBM-roofline case
    if rank == 0 and world == 1 and args.workload == "laplacian4096" and not args.no_extras:
        sh.close()  # the headline plan is done: give its 1 GB back before the other matrices are measured
        del xd, yd
        torch.cuda.empty_cache()
        out["other_workloads"] = {}
        for wl, dt2 in (("scircuit", dtype), ("webbase", dtype), ("nlpkkt160", np.dtype(np.float32))):
            try:
                from tilespmv_amd.tile_matrix import field_array
                import scipy.sparse as sp
                small = wl != "nlpkkt160"
                td2 = torch.float64 if dt2 == np.float64 else torch.float32
                m2, n2, rp2, ci2, src2 = build_matrix(wl)
                r2 = (m2 // 16) * 16; nz2 = int(rp2[r2])
                v2, x2 = G.compat_values(len(ci2), dt2), G.compat_x(n2, dt2)
                # config 2 must exercise all seven tile formats: HYB is only reachable with the opt-in rule
                tm2 = api.Tile_create(r2, n2, nz2, rp2, ci2, v2, dtype=dt2, hyb=(wl == "scircuit"))
                hist = np.bincount(field_array(tm2, "Format", tm2.tilenum), minlength=7).tolist()
                ref2 = sp.csr_matrix((v2[:nz2], ci2[:nz2], rp2[:r2 + 1]), shape=(r2, n2)).astype(np.float64) @ x2.astype(np.float64)
                b2 = api.algorithmic_bytes(nz2, r2, n2, dt2.itemsize)
                rec = {"source": src2, "dtype": "f64" if dt2 == np.float64 else "f32", "rows": r2, "nnz": nz2,
                       "tile_format_histogram[csr,coo,ell,hyb,dns,dnsrow,dnscol]": hist,
                       "note": ("cache-resident: launch/latency-bound, roofline time %.1f us" % (b2 / 8e12 * 1e6)) if small
                               else "HBM-bound; integer-valued data, checked exactly against scipy CSR"}
                xd2 = torch.from_numpy(x2).cuda()
                modes = (("coo_in_tile", api.COO_IN_TILE), ("coo_csr_fallback", api.COO_FALLBACK)) if small else (("default_plan", api.COO_AUTO),)
                for label, coo in modes:
                    p2 = api.Plan(tm2, r2, n2, nz2, coo_mode=coo)
                    yd2 = torch.zeros(r2 + 16, dtype=td2, device="cuda")
                    ms2 = p2.time(xd2.data_ptr(), yd2.data_ptr(), stream.cuda_stream, warmup=20, reps=200 if small else 50)
                    ok2 = bool(np.array_equal(yd2.cpu().numpy()[:r2].astype(np.float64), ref2))
                    rec[label] = {"ms_per_spmv": round(ms2, 5), "gflops": round(2.0 * nz2 / ms2 * 1e-6, 1),
                                  "hbm_gbps_algorithmic": round(b2 / ms2 * 1e-6, 1), "frac_of_8TBps": round(b2 / ms2 * 1e-6 / HBM_PEAK_GBPS, 4),
                                  "check": "pass" if ok2 else "FAIL", "fallback_nnz": p2.info()["fallback_nnz"]}
                    p2.close()
                    del yd2
                out["other_workloads"][wl] = rec
                api.Tile_destroy(tm2)
                del m2, n2, rp2, ci2, v2, x2, ref2, xd2
            except Exception as e:  # never let an extra break the headline line
                out["other_workloads"][wl] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(rows, n, rowptr, colidx, vals, x, dtype)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    sh.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
