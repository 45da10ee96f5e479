#!/usr/bin/env python3
"""bench.py — y = A*x throughput of the HIP TileSpMV engine on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--workload laplacian4096] [--dtype f64]

A "step" is one SpMV over the whole (row-partitioned) matrix.  Inputs are resident in HBM when
the timed region starts.  Rank 0 prints ONE JSON line.  Metric/config follow BASELINE.json:
fp64 SpMV GFLOP/s + achieved (algorithmic) HBM GB/s as a fraction of the 8 TB/s roofline.

Workloads (synthetic stand-ins unless $TILESPMV_MATRIX_DIR/<name>.mtx exists):
  laplacian4096  5-pt Laplacian on a 4096^2 grid, 16.7 M rows, 83.9 M nnz  (config 4; default —
                 the >= 10 M-nnz fp64 case the roofline target is quoted on, fits one GPU)
  scircuit       circuit-like, 170,998 rows, 958,936 nnz (config 2)   webbase   power-law, 1,000,005 rows, 3,105,536 nnz (config 3)
  nlpkkt160      KKT-like, 8,345,600 rows, 229,518,112 nnz, fp32 by default (config 5)
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
    # stand-ins sized like the matrices they stand for (reference src/external/CSR5_cuda/2757-matrix.csv:544,:2379,:1903)
    if name == "scircuit":
        return G.retarget_nnz(*G.circuit_like(170998, seed=1), target_nnz=958936, seed=1) + ("synthetic circuit-like stand-in for scircuit, same rows / nnz",)
    if name == "webbase":
        return G.retarget_nnz(*G.powerlaw(1000005, seed=2), target_nnz=3105536, seed=2) + ("synthetic power-law stand-in for webbase-1M, same rows / nnz",)
    if name == "nlpkkt160":
        return G.nlpkkt_like(160) + ("synthetic KKT stand-in for nlpkkt160, same rows / nnz",)
    raise SystemExit("unknown workload " + name)


def cpu_baseline(rows, cols, rowptr, colidx, vals, x, dtype):
    """The CPU path of the reference timed on this box's host cores (rank 0, N=1 only) on the FULL
    workload, with SURVEY.md S8(d)'s protocol: 2 warm-ups, median of 10.
      value / kind "reference": the reference's own tilespmv_cpu (oracle/_ref, its headers compiled in
        place) — one call = schedule analysis + serial format loop + self-check (it cannot be cut apart
        without editing it), single-threaded like the reference (src/tilespmv_cpu.h:125);
      format_loop_only: our C restatement's serial format loop alone (schedule construction timed
        separately), and the same loop with OpenMP over tile-rows on all host cores.
    The Tile_matrix handed to them is made by the product's Tile_create (byte-identical to the
    reference's, tests/test_host.py): the reference's own is O(tilem x tilen) and does not finish at
    this size.  Falls back to kind "port" when oracle/_ref is absent."""
    from oracle.oracle import CpuImpl, available
    from tilespmv_amd import api
    import ctypes as C
    have_ref = available("ref", dtype)
    O = CpuImpl("oracle", dtype)
    R = CpuImpl("ref", dtype) if have_ref else None
    nz = int(rowptr[rows])
    rp, ci, v = rowptr[:rows + 1], colidx[:nz], vals[:nz]
    devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); sys.stdout.flush(); os.dup2(devnull, 1)
    try:
        tm = api.Tile_create(rows, cols, nz, rp, ci, v, dtype=dtype)   # same struct layout as the checkers' (tests/test_abi.py)
        yg = O.csr_spmv(rows, rp, ci, v, x)

        def med(f, warm=2, runs=10):
            for _ in range(warm):
                f()
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts))

        # raw ctypes calls, buffers allocated once: no wrapper copies inside the timed calls
        n = tm.tilenum
        p1 = np.zeros(max(n, 1), dtype=np.int32); p2 = np.zeros(max(n, 1), dtype=np.int32)
        y = np.zeros(rows + 16, dtype=dtype)
        xs = np.ascontiguousarray(x, dtype=dtype); ygs = np.ascontiguousarray(yg, dtype=dtype)
        rps = np.ascontiguousarray(rp, dtype=np.int32); cis = np.ascontiguousarray(ci, dtype=np.int32); vs = np.ascontiguousarray(v, dtype=dtype)
        VP = C.POINTER(O.vt); IP = C.POINTER(C.c_int); UP = C.POINTER(C.c_uint)
        t_ref = None
        if R:
            fr = R.lib.ref_tilespmv_cpu
            fr.argtypes = [C.POINTER(R.TM), IP, IP, IP, C.POINTER(UP), C.POINTER(IP), C.POINTER(IP), C.c_int, C.c_int, C.c_int, IP, IP, VP, VP, VP, VP]
            fr.restype = None
            def _ref():
                nb = C.c_int(0); a, b, c = UP(), IP(), IP()
                fr(C.byref(tm), p1.ctypes.data_as(IP), p2.ctypes.data_as(IP), C.byref(nb), C.byref(a), C.byref(b), C.byref(c), rows, cols, nz,
                   rps.ctypes.data_as(IP), cis.ctypes.data_as(IP), vs.ctypes.data_as(VP), xs.ctypes.data_as(VP), y.ctypes.data_as(VP), ygs.ctypes.data_as(VP))
                for q in (a, b, c):
                    R.free(C.cast(q, C.c_void_p))
            t_ref = med(_ref)
            ref_ok = bool(np.array_equal(y[:rows], ygs[:rows]))
        f = O.lib.oracle_tilespmv_cpu
        f.argtypes = [C.POINTER(O.TM), IP, IP, C.c_int, C.c_int, VP, VP, VP]; f.restype = C.c_int
        tmO = C.cast(C.byref(tm), C.POINTER(O.TM))
        errs = []
        t_loop = med(lambda: errs.append(f(tmO, p1.ctypes.data_as(IP), p2.ctypes.data_as(IP), rows, cols, xs.ctypes.data_as(VP),
                                           y.ctypes.data_as(VP), ygs.ctypes.data_as(VP))))
        sch = O.lib.oracle_schedule
        sch.argtypes = [C.POINTER(O.TM), C.POINTER(UP), C.POINTER(IP), C.POINTER(IP)]; sch.restype = C.c_int
        def _sched():
            a, b, c = UP(), IP(), IP()
            sch(tmO, C.byref(a), C.byref(b), C.byref(c))
            for q in (a, b, c):
                O.free(C.cast(q, C.c_void_p))
        t_sched = med(_sched, warm=1, runs=3)
        fo = O.lib.oracle_tilespmv_cpu_omp
        fo.argtypes = [C.POINTER(O.TM), C.c_int, C.c_int, VP, VP]; fo.restype = C.c_int
        nthr = []
        t_omp = med(lambda: nthr.append(fo(tmO, rows, cols, xs.ctypes.data_as(VP), y.ctypes.data_as(VP))))
        api.Tile_destroy(tm)
    finally:
        C.CDLL(None).fflush(None)   # the checkers print through C stdio: drain it into /dev/null before stdout comes back
        sys.stdout.flush(); os.dup2(saved, 1); os.close(devnull); os.close(saved)
    gf = lambda t: round(2.0 * nz / t * 1e-9, 4)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    t_main = t_ref if t_ref is not None else t_loop
    return {"value": gf(t_main), "unit": "GFLOP/s", "cores": 1, "kind": "reference" if t_ref is not None else "port",
            "seconds_per_spmv": round(t_main, 6),
            "sample": "the full workload (%d rows, %d nnz); %s; 2 warm-ups, median of 10" % (
                rows, nz, "one tilespmv_cpu call = schedule + serial format loop + self-check" if t_ref is not None else "serial format loop of the restatement"),
            "format_loop_only": {"kind": "port", "serial_1_core": {"value": gf(t_loop), "seconds": round(t_loop, 6), "errcount": int(errs[-1])},
                                 "all_host_cores": {"value": gf(t_omp), "seconds": round(t_omp, 6), "cores": int(nthr[-1])},
                                 "schedule_construction_seconds": round(t_sched, 6), "unit": "GFLOP/s"},
            "reference_y_equals_csr_golden": (ref_ok if t_ref is not None else None),
            "host_cpu": cpu, "host_logical_cores": os.cpu_count()}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args_list, n):
    """`python bench.py --gpus N` started bare (no torch.distributed.run around it): start the N ranks
    as a CHILD process group — before this process has imported torch or touched HIP, and never by exec
    — relay rank 0's JSON line and the children's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["TILESPMV_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + args_list
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    for l in res.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif res.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 1
    return res.returncode


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
    ap.add_argument("--setup-launches", type=int, default=200,
                    help="untimed SpMVs issued during setup, before the --warmup steps (the reference warms up with 200 launches, "
                         "src/tilespmv_cuda.h:1059-1082): clocks and caches are at steady state even when the driver asks for few steps")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo only to rehearse the N>1 path on a single GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

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
        torch.cuda.synchronize()
        own = time.perf_counter() - t0        # this rank's K steps are done (barrier + synchronize before, synchronize here)
        sync_all()                            # ... and the closing barrier + synchronize
        dev_ms = e0.elapsed_time(e1)          # HIP events on the launch stream, over the timed region
        per_rank = [[float(dev), own * 1e3 / steps, dev_ms / steps]]
        wall = own
        if world > 1:
            cdev = "cuda" if args.backend == "nccl" else "cpu"
            mine = torch.tensor(per_rank[0], dtype=torch.float64, device=cdev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)          # every rank's device id, wall ms/step, device ms/step
            per_rank = [[float(v) for v in t.cpu()] for t in allr]
            wall = max(r[1] for r in per_rank) * steps * 1e-3   # MAX over ranks of the time each rank needed for its K steps
            dev_ms = max(r[2] for r in per_rank) * steps
        timed.per_rank = per_rank
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

    if args.setup_launches > 0:   # setup, not part of the W warm-up steps nor of the K timed ones; reported as `setup_launches`
        run("none", args.setup_launches)
        sync_all()
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
    main_per_rank = timed.per_rank
    # HBM traffic of one launch: PMC counters cannot be read inside this process (rocprofv3 wraps the command, and a
    # --pmc pass serialises the kernels), so the figure comes from the committed summary of scripts/profile_round.sh
    # for this workload/dtype and is labelled with its source; null when no such pass has been committed.
    traffic, traffic_source = None, None
    tj = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (args.workload, "f64" if dtype == np.float64 else "f32"))
    if world == 1 and os.path.exists(tj):
        tjd = json.load(open(tj))
        traffic = tjd.get("hbm_bytes_per_launch")
        traffic_source = {"file": os.path.relpath(tj, ROOT), "measured": tjd.get("measured", "round 1"), "kernel": tjd.get("kernel"),
                          "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this command; FETCH_SIZE x2 (gfx950 correction, calibrated)",
                          "live": False}
    # the reference's own timing protocol (src/tilespmv_cuda.h:1112-1137): wall clock around launch + sync, one SpMV at a time
    # (its cudaMemset of y is not needed: the kernel overwrites y)
    nref = min(1000, max(args.steps, 50))
    sync_all()
    tr = []
    for _ in range(nref):
        t1 = time.perf_counter()
        sh.spmv(xd, yd, stream.cuda_stream)
        torch.cuda.synchronize()
        tr.append(time.perf_counter() - t1)
    ref_style_ms = float(np.mean(tr)) * 1e3
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
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
        "setup_launches": args.setup_launches, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64" if dtype == np.float64 else "f32", "data": "synthetic (reference driver data: val[i]=i%10, x[i]=i%10)",
        "config": {"workload": args.workload, "source": source, "rows": rows, "cols": n, "nnz": nnz,
                   "partition": "tile-row blocks, nnz-balanced, %d rank(s)" % world, "y_combine": args.combine, "backend": args.backend if world > 1 else None,
                   "tiles": info["tiles"], "tasks": info["num_tasks"], "coo_mode": info["coo_mode"], "dense_mode": info["dense_mode"],
                   "entry_mode": info["entry_mode"], "sums_bit_reproducible": bool(info["entry_ordered"]), "strip_cost": info["strip_cost"]},
        "hbm_gbps_algorithmic": round(b_alg_total / (wall / args.steps) * 1e-9, 1),
        "hbm_roofline_frac": round(b_alg_total / (wall / args.steps) * 1e-9 / (HBM_PEAK_GBPS * world), 4),
        "roofline": roofline, "check": check,
        "ranks": (dist.get_world_size() if world > 1 else 1), "devices": [int(r[0]) for r in main_per_rank], "backend": (args.backend if world > 1 else None),
        "per_rank_ms_per_step": {"wall": [round(r[1], 5) for r in main_per_rank], "device": [round(r[2], 5) for r in main_per_rank],
                                 "min": round(min(r[1] for r in main_per_rank), 5), "max": round(max(r[1] for r in main_per_rank), 5)},
        "launched_by": "self (child torch.distributed.run)" if os.environ.get("TILESPMV_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "direct"),
        "reference_style_timing": {"ms_per_spmv": round(ref_style_ms, 5), "gflops": round(flops / (ref_style_ms * 1e-3) * 1e-9, 2), "reps": nref,
                                   "protocol": "wall clock around launch + synchronize, one SpMV at a time (reference src/tilespmv_cuda.h:1112-1137), this rank's shard"},
        "prep_seconds": dict({"generate": round(t_gen, 3), "total_tile_create_plus_plan": round(t_prep, 3)},
                             **{k: round(v, 3) for k, v in sh.seconds.items()}),
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
                                  "check": "pass" if ok2 else "FAIL", "fallback_nnz": p2.info()["fallback_nnz"],
                                  "entry_mode": p2.info()["entry_mode"], "sums_bit_reproducible": bool(p2.info()["entry_ordered"]),
                                  "strip_cost": p2.info()["strip_cost"], "tasks": p2.info()["num_tasks"]}
                    p2.close()
                    del yd2
                tj2 = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (wl, "f64" if dt2 == np.float64 else "f32"))
                if os.path.exists(tj2):   # HBM-side bytes per launch of the default plan, from the committed counter passes (not live)
                    t2 = json.load(open(tj2))
                    rec["traffic"] = {"hbm_bytes_per_launch": t2.get("hbm_bytes_per_launch"), "kernel": t2.get("kernel"), "measured": t2.get("measured"),
                                      "file": os.path.relpath(tj2, ROOT), "live": False}
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
