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
  path/to/A.mtx  any Matrix Market file (e.g. a SuiteSparse download), through the product's reader
Multi-GPU: contiguous nnz-balanced tile-row blocks, one rank per GPU, x replicated, y left
sharded (the SpMV needs no collective: SURVEY.md §8e) => "scaling": "strong" on the fixed matrix;
--combine allgather|allreduce adds the RCCL y combine to every step, --combine halo runs the sharded-x form
(HaloSpMV: only the halo of x travels); the default run also reports all three beside the headline value when N > 1.
Every rank generates (or reads) and preprocesses ONLY its own row block where the workload has a row-block generator
(the stencil workloads, incl. the headline) — `generated` in the line says which.
--data real: vals, x ~ U(-1, 1), seed 12345 (SURVEY.md S8(d) data mode ii), whole-y tolerance check instead of the exact one.
--cache DIR: parse / tile once — .mtx files through the CSR cache (mmio_allinone_cached), Tile_matrix through tilespmv_matrix_save/load.
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
SCATTERED_GATHER_CEILING = 59.4e9  # scattered 8-byte gathers per second this chip resolves from a 64-MB table (128 B each at the fabric): scripts/micro/gather_granule.hip, profiles/r04_gather_granule.txt


def build_matrix(name, cache_dir=None):
    from tilespmv_amd import api, generators as G
    if name.endswith(".mtx"):   # any Matrix Market file (a SuiteSparse download): --workload path/to/A.mtx — parsed by the product's reader (file order, symmetric entries mirrored: src/mmio_highlevel.h:593-759), cached when --cache is given
        if not os.path.exists(name):
            raise SystemExit("bench.py: no such file: " + name)
        cache = os.path.join(cache_dir, os.path.basename(name)[:-4] + ".csr_f64") if cache_dir else None
        r = api.mmio_allinone(name, cache=cache)
        how = "" if cache is None else (" (CSR cache hit)" if r.get("from_cache") == 1 else " (parsed; CSR cache written)")
        return r["m"], r["n"], r["rowptr"], r["colidx"], "file:" + os.path.basename(name) + how
    d = os.environ.get("TILESPMV_MATRIX_DIR")
    real = {"laplacian4096": None, "scircuit": "scircuit", "webbase": "webbase-1M", "nlpkkt160": "nlpkkt160"}.get(name)
    if d and real and os.path.exists(os.path.join(d, real + ".mtx")):
        path = os.path.join(d, real + ".mtx")
        cache = os.path.join(cache_dir, real + ".csr_f64") if cache_dir else None   # parse once
        r = api.mmio_allinone(path, cache=cache)
        how = "" if cache is None else (" (CSR cache hit)" if r.get("from_cache") == 1 else " (parsed; CSR cache written)")
        return r["m"], r["n"], r["rowptr"], r["colidx"], "file:" + real + ".mtx" + how
    if name == "laplacian4096":
        return G.laplacian5pt(4096) + ("synthetic 5-pt Laplacian 4096^2",)
    if name.startswith("lap3d"):
        return G.laplacian7pt(int(name[5:])) + ("synthetic 7-pt Laplacian on a cube",)
    if name.startswith("powerlaw"):
        return G.powerlaw(int(name[8:]), seed=2) + ("synthetic power-law",)
    if name.startswith("shell"):   # shell<dof>[s<window>]_<n>: 9-point quadrilateral mesh on n^2 nodes (a shell model: af_shell-like)
        dof, win, nn = parse_fem("fem" + name[5:])
        return G.fem_hex(nn, nn, 1, dof, shuffle=win) + ("synthetic shell-like: 9-point quad mesh %d^2 nodes x %d dof%s" % (nn, dof, ", nodes shuffled in windows of %d" % win if win else ", natural order"),)
    if name.startswith("fem"):   # fem<dof>[s<window>]_<n>: 27-point hexahedral mesh on n^3 nodes, <dof> unknowns per node (natural order, or nodes shuffled inside windows)
        dof, win, nn = parse_fem(name)
        return G.fem_hex(nn, nn, nn, dof, shuffle=win) + ("synthetic FEM-like: 27-point hex mesh %d^3 nodes x %d dof%s" % (nn, dof, ", nodes shuffled in windows of %d" % win if win else ", natural order"),)
    if name.startswith("circuit") and name[7:].isdigit():
        return G.circuit_like(int(name[7:]), seed=1) + ("synthetic circuit-like",)
    if name.startswith("laplacian"):
        return G.laplacian5pt(int(name[len("laplacian"):])) + ("synthetic 5-pt Laplacian",)
    # meshes of the population sweep (round 5): tri<n>[s<w>] 2-D triangulation on n^2 nodes, tet<n>[s<w>] 3-D tetrahedral mesh on n^3 nodes, road<n>[s<w>] road-network-like
    # grid graph on n^2 junctions — natural order, or nodes shuffled inside windows of w; plaw<alpha x 10>_<rows>: power-law with another exponent
    for pre, gen, what in (("tri", lambda k, w: G.tri_mesh(k, k, shuffle=w), "2-D triangulation %d^2 nodes"), ("tet", lambda k, w: G.tet_mesh(k, shuffle=w), "3-D tetrahedral mesh %d^3 nodes"),
                           ("road", lambda k, w: G.road_like(k, k, shuffle=w), "road-network-like grid graph %d^2 junctions")):
        if name.startswith(pre) and name[len(pre):len(pre) + 1].isdigit():
            k, _, w = name[len(pre):].partition("s")
            return gen(int(k), int(w or 0)) + ("synthetic " + what % int(k) + (", nodes shuffled in windows of %s" % w if w else ", natural order"),)
    if name.startswith("plaw"):
        a, nn = name[4:].split("_")
        return G.powerlaw(int(nn), seed=2, alpha=int(a) / 10.0) + ("synthetic power-law, exponent %.1f" % (int(a) / 10.0),)
    # the HBM-resident irregular / mixed class: bandrand<hbw>x<extra>_<rows>, uniform<per_row>_<rows>[x<cols>], rmat<scale>x<edge factor>
    if name.startswith("bandrand"):
        a, nn = name[8:].split("_"); hbw, extra = a.split("x")
        return G.band_plus_random(int(nn), int(hbw), int(extra), 5) + ("synthetic band hbw=%s + %s random entries per row" % (hbw, extra),)
    if name.startswith("uniform"):
        k, nn = name[7:].split("_"); rr, _, cc = nn.partition("x")
        return G.uniform_per_row(int(rr), int(cc or rr), int(k), 1) + ("synthetic uniform random, %s per row" % k,)
    if name.startswith("rmat"):
        sc, ef = name[4:].split("x")
        return G.rmat(int(sc), int(ef), 3) + ("synthetic R-MAT scale %s, %s edges per vertex" % (sc, ef),)
    if name.startswith("blockdiag"):   # blockdiag<bs>x<extra>_<blocks>
        a, nb = name[9:].split("_"); bs, extra = a.split("x")
        return G.block_diag_plus_sparse(int(nb), int(bs), int(extra), 6) + ("synthetic block-diagonal %sx%s (60 %%) + %s random per row" % (bs, bs, extra),)
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


def parse_fem(name):
    a, nn = name[3:].split("_")
    dof, _, win = a.partition("s")
    return int(dof), int(win or 0), int(nn)


def build_block(name, rank, world, cache_dir=None):
    """This rank's row block of the workload: (rows, cols, bounds, rowptr_block, colidx_block, first_nnz, nnz_total, source, how).
    Stencil workloads (incl. the headline) have row-block generators: the cheap row pointer is computed by everyone, the
    nnz-balanced partition cut from it, and only rows [r0, r1) are ever materialised.  Everything else is generated / read
    whole and sliced (`how` says which)."""
    from tilespmv_amd import generators as G
    from tilespmv_amd.dist import partition_rows, shard_csr
    gen = None
    if name.startswith("laplacian") and name[len("laplacian"):].isdigit() and not os.environ.get("TILESPMV_MATRIX_DIR"):
        n = int(name[len("laplacian"):]); gen = (G.laplacian5pt, G.laplacian5pt_rowptr, n, "synthetic 5-pt Laplacian %d^2" % n)
    elif name.startswith("lap3d"):
        n = int(name[5:]); gen = (G.laplacian7pt, G.laplacian7pt_rowptr, n, "synthetic 7-pt Laplacian on a cube")
    elif name.startswith("fem"):
        dof, win, nn = parse_fem(name)
        gen = (lambda n_, rows=None: G.fem_hex(n_, n_, n_, dof, shuffle=win, rows=rows), lambda n_: G.fem_hex(n_, n_, n_, dof, shuffle=win, rowptr_only=True), nn,
               "synthetic FEM-like: 27-point hex mesh %d^3 nodes x %d dof%s" % (nn, dof, ", nodes shuffled in windows of %d" % win if win else ", natural order"))
    if gen is not None:
        full_rp = gen[1](gen[2])
        m = len(full_rp) - 1
        rows = (m // 16) * 16
        bounds = partition_rows(full_rp, rows, world)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        _, cols, rp, ci = gen[0](gen[2], rows=(r0, r1))
        return rows, cols, bounds, rp, ci, int(full_rp[r0]), int(full_rp[rows]), gen[3], "own row block only"
    m, n, rowptr, colidx, source = build_matrix(name, cache_dir)
    rows = (m // 16) * 16
    bounds = partition_rows(rowptr, rows, world)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, _ = shard_csr(rowptr, colidx, colidx, r0, r1)
    return rows, n, bounds, rp, np.ascontiguousarray(ci), int(rowptr[r0]), int(rowptr[rows]), source, "whole matrix, then sliced"


def usable_cores():
    """Cores this process may really run on: affinity mask, capped by the cgroup CPU quota (a GPU box hands out a share of
    the host — omp_get_max_threads() still reports every hardware thread)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return max(1, n)


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
        # all-cores figure: as many threads as this process may really use (affinity + cgroup quota), spread over the cores
        # (proc_bind(spread) clause in the loop), x and y in buffers first touched by those threads
        fo = O.lib.oracle_tilespmv_cpu_omp
        fo.argtypes = [C.POINTER(O.TM), C.c_int, C.c_int, VP, VP]; fo.restype = C.c_int
        O.lib.oracle_set_omp_threads.argtypes = [C.c_int]; O.lib.oracle_set_omp_threads.restype = None
        O.lib.oracle_alloc_first_touch.argtypes = [C.c_size_t]; O.lib.oracle_alloc_first_touch.restype = C.c_void_p
        ncores = usable_cores()
        O.lib.oracle_set_omp_threads(ncores)
        isz = np.dtype(dtype).itemsize
        px, py = O.lib.oracle_alloc_first_touch((cols + 16) * isz), O.lib.oracle_alloc_first_touch((rows + 16) * isz)
        C.memmove(px, xs.ctypes.data, cols * isz)
        nthr = []
        t_omp = med(lambda: nthr.append(fo(tmO, rows, cols, C.cast(px, VP), C.cast(py, VP))))
        omp_ok = bool(np.array_equal(np.frombuffer((C.c_char * (rows * isz)).from_address(py), dtype=dtype), ygs[:rows]))
        O.free(C.c_void_p(px)); O.free(C.c_void_p(py))
        O.lib.oracle_set_omp_threads(0)
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
                                 "all_host_cores": {"value": gf(t_omp), "seconds": round(t_omp, 6), "cores": int(nthr[-1]), "threads": int(nthr[-1]),
                                                    "usable_cores": ncores, "binding": "proc_bind(spread) on the loop; x, y first touched by the worker threads",
                                                    "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
                                                    "y_equals_csr_golden": omp_ok},
                                 "schedule_construction_seconds": round(t_sched, 6), "unit": "GFLOP/s"},
            "reference_y_equals_csr_golden": (ref_ok if t_ref is not None else None),
            "host_cpu": cpu, "host_logical_cores": os.cpu_count()}


def device_state_under_load(run_async, sync, samples=6, gap=0.04):
    """Clocks / power of the card while the SpMV runs (sysfs reads only, no child process): tells a slow box from a slow kernel.
    `run_async` queues ~0.3 s of back-to-back SpMVs, `sync` waits for them.  None when the files are not readable."""
    import glob
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
    if not cards:
        return None

    def busy(c):
        try:
            return int(open(os.path.join(c, "gpu_busy_percent")).read())
        except (OSError, ValueError):
            return -1
    d = cards[0]

    def starred(name):
        try:
            for line in open(os.path.join(d, name)):
                if line.rstrip().endswith("*"):
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def hw(name, scale):
        for f in glob.glob(os.path.join(d, "hwmon", "*", name)):
            try:
                return round(int(open(f).read()) * scale, 1)
            except (OSError, ValueError):
                pass
        return None
    out = {"sclk_mhz": [], "mclk_mhz": [], "fclk_mhz": [], "power_w": []}
    run_async()
    if len(cards) > 1:       # several cards visible in sysfs: ours is the busy one
        time.sleep(gap)
        d = max(cards, key=busy)
    out["card"] = os.path.basename(os.path.dirname(d))
    for _ in range(samples):
        time.sleep(gap)
        out["sclk_mhz"].append(starred("pp_dpm_sclk")); out["mclk_mhz"].append(starred("pp_dpm_mclk")); out["fclk_mhz"].append(starred("pp_dpm_fclk"))
        out["power_w"].append(hw("power1_input", 1e-6))
    sync()
    out["temp_junction_c"] = hw("temp2_input", 1e-3)
    out["power_cap_w"] = hw("power1_cap", 1e-6)
    out["note"] = "sampled from sysfs while ~0.3 s of back-to-back SpMVs run after the timed region"
    return out


LINE_BUDGET = 4096   # bytes of the final stdout line (the driver parses the LAST line; the reference's own record is one short line: src/tilespmv_cuda.h:1139-1147)


def compact_line(out, full_path):
    """The ONE stdout line: the driver's contract keys, the roofline and CPU baseline of the headline, and name -> [ms, frac] for everything
    else measured in the run.  The complete record (every plan parameter, preparation seconds, counter sources ...) goes to `full_path`."""
    rf, cb = out["roofline"], out.get("cpu_baseline")
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "source", "rows", "cols", "nnz", "partition", "y_combine", "tiles", "x_panel_merge", "x_slice_passes") if k in cfg}
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_min_bytes", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
                                               "min_bytes_per_launch", "plan_stream_bytes_per_launch", "actual_traffic_gbps", "plan_fingerprint")}   # (plan_fingerprint: what scripts/traffic_json.py ties a counter pass to)
    line["roofline"]["traffic_from"] = (rf.get("traffic_source") or {}).get("file")
    line["roofline"]["timing"] = "hip events, launch stream, the K timed steps"
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "seconds_per_spmv", "sample")}
        allc = cb["format_loop_only"]["all_host_cores"]
        line["cpu_baseline"]["all_host_cores"] = {"value": allc["value"], "cores": allc["cores"], "kind": "port"}
        line["cpu_baseline"]["host_cpu"] = cb.get("host_cpu")
    else:
        line["cpu_baseline"] = None
    line["check"] = out["check"]
    if out.get("steady_state"):
        line["steady_state"] = out["steady_state"]
    line["reference_style_ms"] = out["reference_style_timing"]["ms_per_spmv"]
    line["ranks"] = out["ranks"]; line["backend"] = out["backend"]; line["launched_by"] = out["launched_by"]
    if out["n_gpus"] > 1:
        line["rank_devices"] = out.get("rank_devices")
        line["rccl_version"] = out.get("rccl_version")
        line["per_rank_kernel_ms"] = out["per_rank_ms_per_step"]["device"]
        line["per_rank_wall_ms"] = out["per_rank_ms_per_step"]["wall"]
        if out.get("with_y_combine"):
            line["with_y_combine"] = {m: ({"ms": v.get("ms_per_step"), "check": v.get("check_full_y_on_every_rank") or v.get("check_own_rows_on_every_rank")} if "error" not in v
                                          else {"error": v["error"][:120]}) for m, v in out["with_y_combine"].items()}
        line["prep_seconds_max"] = max((p or {}).get("total_tile_create_plus_plan", 0.0) for p in out["prep_seconds_per_rank"])
    else:
        line["prep_seconds"] = out["prep_seconds"].get("total_tile_create_plus_plan")
    if "halo_bytes_per_rank" in out:
        line["halo_bytes_per_rank"] = out["halo_bytes_per_rank"]
    ow = out.get("other_workloads")
    if ow:   # name -> [ms per SpMV, frac of 8 TB/s on the CSR-model bytes, whole-y check]
        def brief(rec):
            if "error" in rec:
                return "error"
            best = rec.get("default_plan") or min((rec[k] for k in ("coo_in_tile", "coo_csr_fallback") if k in rec), key=lambda r: r["ms_per_spmv"])
            return [best["ms_per_spmv"], best["frac_of_8TBps"], best["check"]]
        line["other_workloads"] = {k: brief(v) for k, v in ow.items()}
    pop = out.get("population")
    if pop:
        line["population"] = {"error": pop["error"][:120]} if "error" in pop else {
            "this_run": {k: pop["this_run"][k] for k in ("count", "frac_median", "frac_min", "share_frac_ge_0.70")} if "this_run" in pop else None,
            "live": {k: [v.get("ms_per_spmv"), v.get("frac")] for k, v in pop.get("live_subset", {}).items() if isinstance(v, dict)},
            "committed_sweep": {k: pop["committed_sweep"].get(k) for k in ("count", "frac_median", "share_frac_ge_0.70", "file")} if "committed_sweep" in pop else None}
    ro = out.get("reorder")
    if ro and "workloads" in ro:   # name -> [frac natural numbering, frac after RCM (kernel only), frac amortised over 10 products]
        line["reorder"] = {k: ("error" if "error" in v else [v["natural"]["frac"], v["rcm"]["frac"], v["amortised_over_K_products"]["10"]["frac"]]) for k, v in ro["workloads"].items()}
    line["full"] = full_path
    txt = json.dumps(line, separators=(",", ":"))
    for drop in ("reorder", "population", "other_workloads", "steady_state", "with_y_combine"):   # never let the line outgrow what the driver reads: shed the side tables first
        if len(txt) <= LINE_BUDGET:
            break
        if drop in line:
            line[drop] = "see " + str(full_path)
            txt = json.dumps(line, separators=(",", ":"))
    return txt


def write_full(out, path):
    """The complete record beside the line; a read-only checkout must not break the run."""
    for p in (path, os.path.join("/tmp", os.path.basename(path))):
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
            return p
        except OSError:
            continue
    return None


class phase:
    """Hard limit on one phase of the run: when it expires the rank says where it was stuck on stderr and leaves with a non-zero exit code (the launcher then ends the
    other ranks) — a first-contact hang on an 8-GPU node becomes a diagnosis instead of the driver's own timeout.  Nothing is re-executed."""
    scale = 1.0
    log = []

    def __init__(self, name, seconds):
        self.name, self.seconds = name, seconds * phase.scale

    def _expire(self):
        sys.stderr.write("bench.py: rank %s: phase '%s' exceeded %.0f s — giving up (exit 124)\n" % (os.environ.get("RANK", "0"), self.name, self.seconds))
        sys.stderr.flush()
        os._exit(124)

    def __enter__(self):
        import threading
        self.t0 = time.time()
        self.timer = threading.Timer(self.seconds, self._expire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        phase.log.append((self.name, round(time.time() - self.t0, 2)))
        return False


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
    limit = float(os.environ.get("TILESPMV_BENCH_JOB_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)   # a fresh child (its own process group), never an exec
    try:
        stdout, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)     # exactly the group started above
        stdout, _ = proc.communicate()
        print("bench.py: the %d ranks did not finish within %.0f s — killed" % (n, limit), file=sys.stderr)
        return 124
    res = subprocess.CompletedProcess(cmd, proc.returncode, stdout)
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
    ap.add_argument("--combine", default="none", choices=["none", "allgather", "allreduce", "halo"])
    ap.add_argument("--data", default="compat", choices=["compat", "real"],
                    help="compat: the reference driver's val[i] = i %% 10, x[i] = i %% 10 (exact check); real: U(-1, 1), seed 12345 (tolerance check)")
    ap.add_argument("--cache", default=None, metavar="DIR", help="parse / tile once: CSR cache of .mtx inputs and Tile_matrix cache of this rank's block live here")
    ap.add_argument("--prep", default="host", choices=["host", "device"],
                    help="how the measured plan is prepared: Tile_create on the host + tilespmv_plan_create (the reference's flow), or tilespmv_plan_create_from_csr (tiled matrix and plan built on the device); "
                         "either way `prep_seconds` reports both")
    ap.add_argument("--no-prep-warm-up", action="store_true",
                    help="time the preparation in a cold process (by default a small matrix goes through both preparation paths first, untimed: `process_warm_up_seconds`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras", default=None, help="comma-separated keys of `other_workloads` to run (default: all)")
    ap.add_argument("--setup-launches", type=int, default=200,
                    help="untimed SpMVs issued before the steady-state measurement (the reference warms up with 200 launches, "
                         "src/tilespmv_cuda.h:1059-1082).  The plain protocol (W warm-ups, K steps, nothing else) is measured FIRST and reported beside it")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"), metavar="PATH",
                    help="the complete record (every plan parameter, preparation seconds, per-workload detail) is written here; stdout carries one compact line (< 4 KB) that names this file")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one GPU per rank); gloo only to rehearse the N>1 path on a single GPU")
    ap.add_argument("--phase-timeout-scale", type=float, default=1.0, help="multiplies the hard per-phase limits (init 300 s, prepare 900 s, check 300 s, timed 300 s, each combine 300 s, extras 1500 s, cpu baseline 600 s)")
    args = ap.parse_args()
    phase.scale = args.phase_timeout_scale
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything touches the GPU: this pool's host driver only supports dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument without it)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from tilespmv_amd import api, generators as G
    from tilespmv_amd.dist import ShardedSpMV

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:  # one process per GPU shares the host cores: this rank's share of what the job may really use (affinity mask and cgroup quota, not os.cpu_count())
        os.environ.setdefault("TILESPMV_NUM_THREADS", str(max(1, min(16, usable_cores() // world))))
    host_threads = int(os.environ.get("TILESPMV_NUM_THREADS", "0")) or None
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    rccl_version = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with phase("init_process_group", 300):
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
                try:
                    rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception as e:
                    rccl_version = "unknown (%r)" % (e,)
                probe = torch.ones(1, device="cuda")       # first contact: one tiny all-reduce before any real work, so a fabric / IPC problem shows up here, by name
                dist.all_reduce(probe); torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise SystemExit("bench.py: first RCCL all_reduce returned %r, expected %d" % (probe.item(), world))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
    # who runs where: every rank's device as the runtime names it (two ranks on one card, or a card seen twice, is visible in the line)
    pr_ = torch.cuda.get_device_properties(dev)
    my_dev = {"rank": rank, "device": dev, "name": pr_.name, "uuid": str(getattr(pr_, "uuid", "")),
              "pci": "%04x:%02x:%02x" % (getattr(pr_, "pci_domain_id", 0), getattr(pr_, "pci_bus_id", 0), getattr(pr_, "pci_device_id", 0)), "cus": pr_.multi_processor_count}
    rank_devices = [my_dev]
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, my_dev)
    if args.cache:
        os.makedirs(args.cache, exist_ok=True)

    dtype = np.dtype(np.float32 if (args.dtype == "f32" or (args.dtype is None and args.workload == "nlpkkt160")) else np.float64)
    tdtype = torch.float64 if dtype == np.float64 else torch.float32
    dname = "f64" if dtype == np.float64 else "f32"
    t0 = time.time()
    rows, n, bounds, rp_b, ci_b, first_nnz, nnz, source, generated = build_block(args.workload, rank, world, args.cache)
    if args.data == "real":   # one global U(-1, 1) stream: this rank's slice of the values, the whole x
        vals_b, x = G.real_values(len(ci_b), dtype, first=first_nnz), G.real_x(n, nnz, dtype)
    else:                     # reference's synthetic data (src/main.cu:68-69,:93-97), i = global nonzero index
        vals_b, x = G.compat_values(len(ci_b), dtype, first=first_nnz), G.compat_x(n, dtype)
    t_gen = time.time() - t0

    # What a process pays ONCE, whatever it prepares first: the code objects of the preparation kernels, HIP's large-copy path (the first hipMemcpy of hundreds of MB from pageable memory takes
    # 150-280 ms in a fresh process, the same copy 19 ms afterwards — scripts/archive/rounds/r5b_upload_first_touch2.py), the host thread pool.  A small matrix goes through both preparation paths here,
    # untimed, so that `prep_seconds` below is what preparing THIS matrix costs a running process; the one-time part is reported as `process_warm_up_seconds`.
    t0 = time.time()
    if not args.no_prep_warm_up:
        wm, wn, wrp, wci = G.laplacian5pt(1536)
        wv = G.compat_values(len(wci), dtype)
        wtm = api.Tile_create(wm, wn, int(wrp[wm]), wrp, wci, wv, dtype=dtype)
        wp = api.Plan(wtm, wm, wn, int(wrp[wm]), placement_tries=1)
        wp.close(); api.Tile_destroy(wtm)
        try:
            wp = api.Plan.from_csr(wm, wn, int(wrp[wm]), wrp, wci, wv, dtype=dtype, placement_tries=1)
            wp.close()
        except RuntimeError:
            pass
        del wrp, wci, wv
    t_warm_process = time.time() - t0

    t0 = time.time()
    tile_cache = os.path.join(args.cache, "%s_%s_%s_rank%dof%d.tile_%s" % (args.workload, args.data, dname, rank, world, dname)) if args.cache else None
    # config 2 must exercise all seven tile formats: HYB is only reachable with the opt-in rule, here as in `other_workloads`
    prep_on_device = args.prep == "device"
    with phase("prepare (Tile_create + plan)", 900):
        sh = ShardedSpMV(rank, world, rows, n, rp_b, ci_b, vals_b, dtype, bounds=bounds, tile_cache=None if prep_on_device else tile_cache, hyb=(args.workload == "scircuit"), device_build=prep_on_device)
    t_prep = time.time() - t0
    info = sh.local.info()
    stream = torch.cuda.current_stream()
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros(rows + 16, dtype=tdtype, device="cuda")
    halo = None

    def make_halo():
        from tilespmv_amd.halo import HaloSpMV
        if rows != n:
            raise SystemExit("--combine halo needs a square matrix whose size is a multiple of 16 (x is sharded like y)")
        h = HaloSpMV(rank, world, rows, rp_b, ci_b, vals_b, dtype, bounds=bounds)
        h.x_own.copy_(xd[h.r0:h.r1])
        h.y_own = h.new_vector()
        return h

    def step(combine):
        if combine == "halo":
            halo.matvec(halo.x_own, halo.y_own)
            return
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

    # parity of the resident plan before timing, on this rank's WHOLE row block: exact for the reference's integer-valued
    # data; for real-valued data |y - y_ref| <= tol * sum_j |a_ij x_j| with tol 1e-12 (fp64) / 1e-5 (fp32) (SURVEY S8(d) ii;
    # the reference's own GPU check is 1 % relative, src/main.cu:186-197)
    import scipy.sparse as sp
    A64 = sp.csr_matrix((vals_b.astype(np.float64), ci_b, rp_b), shape=(sh.r1 - sh.r0, n)) if not args.no_check else None

    def check_rows(got, what):
        want = A64 @ x.astype(np.float64)
        if args.data == "compat":
            ok = bool(np.array_equal(got.astype(np.float64), want))
        else:
            bound = (1e-12 if dtype == np.float64 else 1e-5) * (abs(A64) @ np.abs(x.astype(np.float64)))
            ok = bool(np.all(np.abs(got.astype(np.float64) - want) <= bound))
        if not ok:
            raise SystemExit("bench.py: HIP result (%s) differs from the CSR golden on rank %d" % (what, rank))
        return "pass"

    check = "skipped"
    with phase("parity check", 300):
        if not args.no_check:
            step("none"); torch.cuda.synchronize()
            check = check_rows(yd[sh.r0:sh.r1].cpu().numpy(), "sharded SpMV")
        if args.combine == "halo":
            halo = make_halo()
            if not args.no_check:
                step("halo"); torch.cuda.synchronize()
                check_rows(halo.y_own[:halo.nloc].cpu().numpy(), "halo SpMV")

    # the headline: the asked protocol — W warm-up steps, K timed steps, nothing else before them
    with phase("timed steps", 300):
        wall, dev_ms = timed(args.combine, args.steps, args.warmup)
    main_per_rank = timed.per_rank
    # beside it (never `value`): steady state — `setup_launches` untimed SpMVs first (reference: 200 warm-up launches, src/tilespmv_cuda.h:1059-1082), then the same W + K
    steady = None
    if args.setup_launches > 0:
        run("none", args.setup_launches)
        sync_all()
        wall_s, dev_ms_s = timed(args.combine, args.steps, args.warmup)
        steady = {"setup_launches": args.setup_launches, "value": round(2.0 * nnz / (wall_s / args.steps) * 1e-9, 2), "ms_per_step": round(wall_s * 1e3 / args.steps, 5),
                  "kernel_ms": round(dev_ms_s / args.steps, 5)}
    ms_per_step = wall * 1e3 / args.steps
    flops = 2.0 * nnz
    b_alg_total = api.algorithmic_bytes(nnz, rows, n, dtype.itemsize)
    value = flops / (wall / args.steps) * 1e-9

    # roofline of the dominant kernel (k_units): algorithmic bytes of THIS rank's launch divided by its average duration,
    # from HIP events on the launch stream over the timed region.
    b_alg_launch = api.algorithmic_bytes(sh.local_nnz, sh.local_rows, n, dtype.itemsize)
    kernel_ms = dev_ms / args.steps
    achieved = b_alg_launch / (kernel_ms * 1e-3) * 1e-9
    # HBM traffic of one launch: PMC counters cannot be read inside this process (rocprofv3 wraps the command, and a
    # --pmc pass serialises the kernels), so the figure comes from the committed summary of scripts/profile_traffic.sh
    # for this workload/dtype — and only if that pass saw THIS plan: the summary carries a fingerprint of the plan it
    # measured (entry mode, strip cost, tasks, stream bytes); on a mismatch traffic is null and marked stale.
    traffic, traffic_source = None, None
    FP_KEYS = ("entry_mode", "strip_cost", "num_tasks", "stream_bytes", "desc_bytes", "nt_stream", "x_panels", "x_slice_passes")
    fingerprint = {k: info[k] for k in FP_KEYS}
    tj = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (args.workload, dname))
    if world == 1 and os.path.exists(tj):
        tjd = json.load(open(tj))
        fresh = {k: (tjd.get("plan_fingerprint") or {}).get(k) for k in FP_KEYS} == fingerprint
        traffic = (tjd.get("hbm_bytes_per_spmv") or tjd.get("hbm_bytes_per_launch")) if fresh else None   # (per SpMV: a plan with dense-tile or column-panel passes is several launches)
        traffic_source = {"file": os.path.relpath(tj, ROOT), "measured": tjd.get("measured"), "kernel": tjd.get("kernel"),
                          "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this command; FETCH_SIZE x2 (gfx950 correction, calibrated)",
                          "live": False, "plan_fingerprint_matches": fresh}
        if not fresh:
            traffic_source["stale"] = "the committed pass measured another plan: %s" % json.dumps(tjd.get("plan_fingerprint"))
    # the reference's own timing protocol (src/tilespmv_cuda.h:1112-1137): wall clock around launch + synchronize, one SpMV at
    # a time, as a C loop (tilespmv_plan_time_reference_style); its cudaMemset of y is not needed: the kernel overwrites y
    nref = min(1000, max(args.steps, 50))
    sync_all()
    ref_style_ms = sh.local.time_reference_style(xd.data_ptr(), yd.data_ptr() + sh.r0 * yd.element_size(), stream.cuda_stream, nref)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "k_units" if info["kernel"] == 2 else "k_tiles_direct", "kernel_ms": round(kernel_ms, 5), "algorithmic_bytes_per_launch": int(b_alg_launch),
                "plan_stream_bytes_per_launch": info["stream_bytes"], "plan_fingerprint": fingerprint, "timing": "hip events on the launch stream, timed region"}
    # `achieved` is priced on the CSR-model bytes SURVEY S8(d) defines (nnz (s_v + 4) + 4 (m + 1) + s_v (n + m)); the tiled plan itself moves fewer
    # (4-bit tile-local columns, 4-B unit descriptors), so that rate can pass the pin bandwidth.  The same launch by the plan's own bytes:
    # ... and by the bytes NO lossless format of this matrix can avoid: every value once, x once, y once (s_v nnz + s_v (n + m)) — a fraction that cannot pass 1
    min_bytes_launch = dtype.itemsize * (sh.local_nnz + n + sh.local_rows)
    roofline["min_bytes_per_launch"] = int(min_bytes_launch)
    roofline["frac_min_bytes"] = round(min_bytes_launch / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 4)
    roofline["plan_bytes_gbps"] = round(info["stream_bytes"] / (kernel_ms * 1e-3) * 1e-9, 1)
    roofline["frac_by_plan_bytes"] = round(info["stream_bytes"] / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 4)
    roofline["note"] = ("`frac` prices the launch on SURVEY S8(d)'s CSR-model bytes (kept for continuity); the tiled plan's streams are %.0f %% of them, so `frac` can pass 1. "
                        "Read `frac_min_bytes` (values + x + y only: bounded by 1), `frac_by_plan_bytes` and, when present, actual_traffic_gbps (counter bytes) as efficiencies"
                        % (100.0 * info["stream_bytes"] / b_alg_launch))

    device_state = None
    if rank == 0 and world == 1:
        try:
            n_load = max(200, int(0.3 / max(kernel_ms * 1e-3, 1e-6)))
            device_state = device_state_under_load(lambda: sh.spmv(xd, yd, stream.cuda_stream, count=n_load), sync_all)
        except Exception as e:  # a diagnostic must never break the line
            device_state = {"error": repr(e)}

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
        ksteps = max(5, args.steps // 10)
        for mode in ("allgather", "allreduce", "halo"):
            try:
              with phase("y combine: " + mode, 300):
                if mode == "halo":
                    halo = make_halo()
                elif not args.no_check:
                    # the first rows of every rank's block as that rank computed them, BEFORE any combine (its whole block was checked against the CSR golden above):
                    # what the neighbours' rows of the combined y are compared with below — a rank-consistent misplacement in the combine cannot pass
                    step("none"); torch.cuda.synchronize()
                    mine_first = yd[sh.r0:min(sh.r0 + 2048, sh.r1)].cpu().numpy().copy()
                    first_rows = [None] * world
                    dist.all_gather_object(first_rows, mine_first)
                w2, _ = timed(mode, ksteps, 3)
                extra[mode] = {"ms_per_step": round(w2 * 1e3 / ksteps, 5), "gflops": round(flops / (w2 / ksteps) * 1e-9, 2)}
                if mode == "halo":
                    extra[mode]["halo_bytes_per_rank"] = int(halo.halo_bytes())
                    extra[mode]["note"] = "x sharded like y: one all_to_all_single of the halo per SpMV, no full-length vector anywhere"
                if not args.no_check:
                    if mode == "halo":   # every rank: its own rows, whole block
                        okv = 1.0
                        try:
                            check_rows(halo.y_own[:halo.nloc].cpu().numpy(), "halo SpMV")
                        except SystemExit:
                            okv = 0.0
                        key = "check_own_rows_on_every_rank"
                    else:                # after the combine every rank holds the whole y: its own block again + the neighbours' first rows
                        nb = (rank + 1) % world
                        ra = np.arange(int(bounds[nb]), min(int(bounds[nb]) + 2048, int(bounds[nb + 1])))
                        mine_ok = check_rows(yd[sh.r0:sh.r1].cpu().numpy(), mode) == "pass"
                        got_nb = yd[torch.from_numpy(ra).cuda()].cpu().numpy()       # the next rank's first rows as they arrived here through the combine
                        okv = 1.0 if (mine_ok and np.array_equal(got_nb, first_rows[nb][:len(ra)])) else 0.0
                        key = "check_full_y_on_every_rank"
                    okc = torch.tensor([okv], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                    dist.all_reduce(okc, op=dist.ReduceOp.MIN)
                    extra[mode][key] = "pass" if float(okc[0]) == 1.0 else "FAIL"
            except Exception as e:   # a combine variant never breaks the headline line
                extra[mode] = {"error": repr(e)}

    # per-rank preprocessing seconds (every rank prepares only its own block)
    prep_mine = dict({"generate": round(t_gen, 3), "process_warm_up_seconds": round(t_warm_process, 3), "total_tile_create_plus_plan": round(t_prep, 3)}, **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in sh.seconds.items()})
    # the other way of preparing the same plan, timed beside it: its streams must be the measured plan's, byte for byte (per-stream digests read back from the device)
    if not args.no_extras:
        try:
            t0 = time.time()
            if prep_on_device:
                tm_o = api.Tile_create(len(rp_b) - 1, n, int(rp_b[-1]), rp_b, ci_b, vals_b, dtype=dtype, hyb=(args.workload == "scircuit"))
                t_tc_o = time.time() - t0
                other = api.Plan(tm_o, len(rp_b) - 1, n, int(rp_b[-1]))
                api.Tile_destroy(tm_o)
            else:
                other = api.Plan.from_csr(len(rp_b) - 1, n, int(rp_b[-1]), rp_b, ci_b, vals_b, dtype=dtype, hyb=(args.workload == "scircuit"))
                t_tc_o = other.info()["tile_create_us"] * 1e-6
            t_o = time.time() - t0
            a, b = sh.local.stream_digests(), other.stream_digests()
            same = sorted(a) == sorted(b) and all(a[k] == b[k] for k in a)
            oi = other.info()
            prep_mine["other_path"] = {"built_on": "host" if prep_on_device else "device", "total_tile_create_plus_plan": round(t_o, 3), "tile_create": round(t_tc_o, 3),
                                       "timed_choices_ms": round(oi["timed_choices_us"] * 1e-3, 1),
                                       "streams_identical_to_measured_plan": bool(same) if oi["timed_choices_us"] == 0 and info["timed_choices_us"] == 0 else ("n/a: a timed choice was made" if not same else True)}
            other.close()
        except NotImplementedError:
            prep_mine["other_path"] = "no device path for this option set"

    if getattr(sh, "tile_cache", None):
        prep_mine["tile_cache"] = sh.tile_cache
    prep_all = [prep_mine]
    if world > 1:
        prep_all = [None] * world
        dist.all_gather_object(prep_all, prep_mine)

    out = {
        "metric": "fp%d SpMV GFLOP/s (y = A*x, tiled format)" % (dtype.itemsize * 8), "value": round(value, 2), "unit": "GFLOP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "steady_state": steady,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": dname,
        "data": "synthetic (reference driver data: val[i]=i%10, x[i]=i%10)" if args.data == "compat" else "synthetic (vals, x ~ U(-1,1), seed 12345; SURVEY S8(d) data mode ii)",
        "config": {"workload": args.workload, "source": source, "generated": generated, "rows": rows, "cols": n, "nnz": nnz,
                   "partition": "tile-row blocks, nnz-balanced, %d rank(s)" % world, "y_combine": args.combine, "backend": args.backend if world > 1 else None,
                   "tiles": getattr(sh, "tiles", None), "tasks": info["num_tasks"], "coo_mode": info["coo_mode"], "dense_mode": info["dense_mode"],
                   "entry_mode": info["entry_mode"], "sums_bit_reproducible": bool(info["entry_ordered"]), "strip_cost": info["strip_cost"],
                   "x_panels": info["x_panels"], "x_panel_merge": info["x_panel_merge"], "x_slice_passes": info["x_slice_passes"], "placement_tries": info["placement_tries"],
                   "csr_form": info["csr_form"], "timed_choices_ms": round(info["timed_choices_us"] * 1e-3, 1)},
        "hbm_gbps_algorithmic": round(b_alg_total / (wall / args.steps) * 1e-9, 1),
        "hbm_roofline_frac": round(b_alg_total / (wall / args.steps) * 1e-9 / (HBM_PEAK_GBPS * world), 4),
        "roofline": roofline,
        "device_state_under_load": device_state,
        "check": check if check == "skipped" else ("pass: whole row block of every rank, %s" % ("exact" if args.data == "compat" else "|y - y_ref| <= %g * sum|a_ij x_j|" % (1e-12 if dtype == np.float64 else 1e-5))),
        "ranks": (dist.get_world_size() if world > 1 else 1), "devices": [int(r[0]) for r in main_per_rank], "backend": (args.backend if world > 1 else None),
        "rank_devices": rank_devices, "rccl_version": rccl_version,
        "per_rank_ms_per_step": {"wall": [round(r[1], 5) for r in main_per_rank], "device": [round(r[2], 5) for r in main_per_rank],
                                 "min": round(min(r[1] for r in main_per_rank), 5), "max": round(max(r[1] for r in main_per_rank), 5)},
        "launched_by": "self (child torch.distributed.run)" if os.environ.get("TILESPMV_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "direct"),
        "reference_style_timing": {"ms_per_spmv": round(ref_style_ms, 5), "gflops": round(2.0 * sh.local_nnz / (ref_style_ms * 1e-3) * 1e-9, 2), "reps": nref,
                                   "protocol": "C loop: gettimeofday around one launch + stream synchronize (reference src/tilespmv_cuda.h:1112-1137), this rank's shard"},
        "prep_seconds": prep_mine, "prep_seconds_per_rank": prep_all, "host_threads_per_rank": host_threads, "usable_host_cores": usable_cores(),
    }
    if extra:
        out["with_y_combine"] = extra
    if args.combine == "halo" and halo is not None:
        out["halo_bytes_per_rank"] = int(halo.halo_bytes())
    vals, rowptr, colidx = vals_b, rp_b, ci_b   # (world == 1 below: the block is the whole matrix)
    # the other BASELINE configs (stand-ins) at one GPU, reported beside the headline; never `value`:
    # configs 2-3 are cache-resident (launch/latency-bound), config 5 is the fp32 HBM-roofline case
    if rank == 0 and world == 1 and args.workload == "laplacian4096" and not args.no_extras:
        sh.close()  # the headline plan is done: give its 1 GB back before the other matrices are measured
        del xd, yd
        torch.cuda.empty_cache()
        out["other_workloads"] = {}
        f32_, f64_ = np.dtype(np.float32), np.dtype(np.float64)
        # key, bench workload, dtype, class ("small": cache-resident BASELINE configs, both COO modes; "large": HBM-bound), also on real-valued data?
        # configs 2, 3, 5 first; then what the north star names beside them — synthetic banded / power-law matrices — and the >= 10 M-nnz irregular / mixed class
        specs = [("scircuit", "scircuit", dtype, "small", True), ("webbase", "webbase", dtype, "small", True), ("nlpkkt160", "nlpkkt160", f32_, "large", True),
                 ("nlpkkt160_f64", "nlpkkt160", f64_, "large", False), ("lap3d256", "lap3d256", f64_, "large", False), ("band40_2m", "band40_2000000", f64_, "large", False),
                 ("powerlaw8m", "powerlaw8000000", f64_, "large", False), ("bandrand4x3_2m", "bandrand4x3_2000000", f64_, "large", False),
                 ("uniform8_4m", "uniform8_4000000", f64_, "large", False), ("uniform8_8m", "uniform8_8000000", f64_, "large", False),
                 # round 5: the FEM / block-structured class (the largest group among the >= 10 M-nnz matrices of the reference's sweep list): 27-point hex meshes, 3 and 6 unknowns
                 # per node, natural order and nodes shuffled inside windows of 64 — > 90 % of their nonzeros sit in CSR-format tiles, executed as pooled units
                 ("fem3_68", "fem3_68", f64_, "large", False), ("fem6_46", "fem6_46", f64_, "large", False), ("fem3s64_68", "fem3s64_68", f64_, "large", False)]
        if args.extras:
            keep = set(args.extras.split(","))
            specs = [sp_ for sp_ in specs if sp_[0] in keep]
        built = {}
        t_extras = time.time()
        ph_extras = phase("other workloads + population", 1500).__enter__()
        for key, wl, dt2, klass, also_real in specs:
            try:
                from tilespmv_amd.tile_matrix import field_array
                import scipy.sparse as sp
                t_wl = time.time()
                small = klass == "small"
                td2 = torch.float64 if dt2 == np.float64 else torch.float32
                if wl not in built:
                    built.clear()           # (one structure at a time: the KKT stand-in serves its fp32 and fp64 entries)
                    built[wl] = build_matrix(wl)
                m2, n2, rp2, ci2, src2 = built[wl]
                r2 = (m2 // 16) * 16; nz2 = int(rp2[r2])
                v2, x2 = G.compat_values(len(ci2), dt2), G.compat_x(n2, dt2)
                # config 2 must exercise all seven tile formats: HYB is only reachable with the opt-in rule
                t_tc = time.time()
                tm2 = api.Tile_create(r2, n2, nz2, rp2, ci2, v2, dtype=dt2, hyb=(wl == "scircuit"))
                t_tc = time.time() - t_tc
                hist = np.bincount(field_array(tm2, "Format", tm2.tilenum), minlength=7).tolist()
                ref2 = sp.csr_matrix((v2[:nz2], ci2[:nz2], rp2[:r2 + 1]), shape=(r2, n2)).astype(np.float64) @ x2.astype(np.float64)
                b2 = api.algorithmic_bytes(nz2, r2, n2, dt2.itemsize)
                bmin2 = dt2.itemsize * (nz2 + n2 + r2)          # values + x + y: what no lossless format can avoid
                rec = {"workload": wl, "source": src2, "stand_in": not src2.startswith("file:"), "dtype": "f64" if dt2 == np.float64 else "f32", "rows": r2, "nnz": nz2,
                       "tile_format_histogram[csr,coo,ell,hyb,dns,dnsrow,dnscol]": hist,
                       "algorithmic_bytes": int(b2), "min_bytes": int(bmin2),
                       "note": ("cache-resident: launch/latency-bound, roofline time %.1f us" % (b2 / 8e12 * 1e6)) if small
                               else "HBM-bound; integer-valued data, checked exactly against scipy CSR"}
                xd2 = torch.from_numpy(x2).cuda()
                modes = (("coo_in_tile", api.COO_IN_TILE), ("coo_csr_fallback", api.COO_FALLBACK)) if small else (("default_plan", api.COO_AUTO),)
                rec["tile_create_seconds"] = round(t_tc, 3)
                if True:   # (config 2's HYB tiles included: built on the device since round 6)
                    # the same default plan prepared on the device (tilespmv_plan_create_from_csr: only the CSR arrays cross the bus), whole y checked like the host-built plan's
                    t_dv = time.time()
                    pdv = api.Plan.from_csr(r2, n2, nz2, rp2, ci2, v2, dtype=dt2, hyb=(wl == "scircuit"))
                    t_dv = time.time() - t_dv
                    ydv = torch.zeros(r2 + 16, dtype=td2, device="cuda")
                    pdv.spmv(xd2.data_ptr(), ydv.data_ptr(), stream.cuda_stream); torch.cuda.synchronize()
                    idv = pdv.info()
                    rec["prepared_on_device"] = {"csr_to_plan_seconds": round(t_dv, 3), "tile_create_incl_csr_upload_seconds": round(idv["tile_create_us"] * 1e-6, 3),
                                                 "timed_choices_ms": round(idv["timed_choices_us"] * 1e-3, 1), "check": "pass" if bool(np.array_equal(ydv.cpu().numpy()[:r2].astype(np.float64), ref2)) else "FAIL"}
                    pdv.close(); del ydv
                for label, coo in modes:
                    t_pc = time.time()
                    p2 = api.Plan(tm2, r2, n2, nz2, coo_mode=coo)
                    t_pc = time.time() - t_pc
                    yd2 = torch.zeros(r2 + 16, dtype=td2, device="cuda")
                    ms2 = p2.time(xd2.data_ptr(), yd2.data_ptr(), stream.cuda_stream, warmup=20, reps=200 if small else 50)
                    i2 = p2.info()
                    if label != "coo_csr_fallback":   # the plan the committed counter passes of this workload measured
                        fp2 = {k: i2[k] for k in FP_KEYS}
                    ok2 = bool(np.array_equal(yd2.cpu().numpy()[:r2].astype(np.float64), ref2))
                    rec[label] = {"ms_per_spmv": round(ms2, 5), "gflops": round(2.0 * nz2 / ms2 * 1e-6, 1),
                                  "hbm_gbps_algorithmic": round(b2 / ms2 * 1e-6, 1), "frac_of_8TBps": round(b2 / ms2 * 1e-6 / HBM_PEAK_GBPS, 4),
                                  "frac_min_bytes": round(bmin2 / ms2 * 1e-6 / HBM_PEAK_GBPS, 4), "frac_by_plan_bytes": round(i2["stream_bytes"] / ms2 * 1e-6 / HBM_PEAK_GBPS, 4),
                                  "check": "pass" if ok2 else "FAIL", "fallback_nnz": i2["fallback_nnz"],
                                  "entry_mode": i2["entry_mode"], "sums_bit_reproducible": bool(i2["entry_ordered"]),
                                  "strip_cost": i2["strip_cost"], "tasks": i2["num_tasks"], "x_panels": i2["x_panels"], "x_slice_passes": i2["x_slice_passes"], "nt_stream": i2["nt_stream"],
                                  "placement_tries": i2["placement_tries"], "csr_form": i2["csr_form"],
                                  # what creating this plan cost (host re-layout + upload + whatever was timed on the device), and how much of it went into choices made by a stopwatch
                                  "plan_create_seconds": round(t_pc, 3), "timed_choices_ms": round(i2["timed_choices_us"] * 1e-3, 1)}
                    if i2["timed_choices_us"] > 0 and label != "coo_csr_fallback":
                        # the same matrix with the one documented switch: no timed choice, every sum in a plan-fixed order (tilespmv_plan_options.deterministic) — what reproducibility costs here
                        t_pd = time.time()
                        pd_ = api.Plan(tm2, r2, n2, nz2, coo_mode=coo, deterministic=1)
                        t_pd = time.time() - t_pd
                        msd = pd_.time(xd2.data_ptr(), yd2.data_ptr(), stream.cuda_stream, warmup=20, reps=200 if small else 50)
                        okd = bool(np.array_equal(yd2.cpu().numpy()[:r2].astype(np.float64), ref2))
                        idt = pd_.info()
                        rec[label]["deterministic"] = {"ms_per_spmv": round(msd, 5), "slowdown": round(msd / ms2, 3), "plan_create_seconds": round(t_pd, 3), "check": "pass" if okd else "FAIL",
                                                       "sums_bit_reproducible": bool(idt["entry_ordered"]), "timed_choices_ms": round(idt["timed_choices_us"] * 1e-3, 1)}
                        pd_.close()
                    if i2["scattered_entries"] * 5 >= nz2 and n2 * np.dtype(dt2).itemsize > (8 << 20):   # a scattered matrix whose x is larger than two L2s (every XCD gathers from all of x): the chip's ceiling for gathers that no neighbour shares (59 G/s, profiles/r04_gather_granule.txt) beside the byte roofline
                        rec[label]["scattered_gathers"] = {"count": i2["scattered_entries"], "share_of_nnz": round(i2["scattered_entries"] / nz2, 3),
                                                           "gathers_per_second": round(i2["scattered_entries"] / ms2 * 1e3, 0), "chip_ceiling_unstructured": SCATTERED_GATHER_CEILING,
                                                           "rate_vs_unstructured_ceiling": round(i2["scattered_entries"] / ms2 * 1e3 / SCATTERED_GATHER_CEILING, 3),
                                                           "note": "a ratio, not a roofline fraction: the ceiling is what the chip sustains when every gather misses its L2; above 1 = the matrix's popular columns or the plan's column panels make gathers hit"}
                    if also_real and label in ("coo_in_tile", "default_plan"):   # the same plan on real-valued data: the time does not depend on the values
                        vr, xr = G.real_values(len(ci2), dt2), G.real_x(n2, len(ci2), dt2)
                        tmr = api.Tile_create(r2, n2, nz2, rp2, ci2, vr, dtype=dt2, hyb=(wl == "scircuit"))
                        pr = api.Plan(tmr, r2, n2, nz2, coo_mode=coo)
                        xdr = torch.from_numpy(xr).cuda()
                        msr = pr.time(xdr.data_ptr(), yd2.data_ptr(), stream.cuda_stream, warmup=20, reps=200 if small else 50)
                        Ar = sp.csr_matrix((vr[:nz2].astype(np.float64), ci2[:nz2], rp2[:r2 + 1]), shape=(r2, n2))
                        wantr = Ar @ xr.astype(np.float64)
                        tol = (1e-12 if dt2 == np.float64 else 1e-5) * (abs(Ar) @ np.abs(xr.astype(np.float64)))
                        okr = bool(np.all(np.abs(yd2.cpu().numpy()[:r2].astype(np.float64) - wantr) <= tol))
                        rec[label]["real_valued_data"] = {"ms_per_spmv": round(msr, 5), "check_whole_y_within_tolerance": "pass" if okr else "FAIL"}
                        pr.close(); api.Tile_destroy(tmr)
                        del xdr, vr, xr, Ar, wantr, tol
                    p2.close()
                    del yd2
                tj2 = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (wl, "f64" if dt2 == np.float64 else "f32"))
                if os.path.exists(tj2):   # HBM-side bytes per launch of the default plan, from the committed counter passes (not live)
                    t2 = json.load(open(tj2))
                    fresh2 = {k: (t2.get("plan_fingerprint") or {}).get(k) for k in FP_KEYS} == fp2
                    rec["traffic"] = {"hbm_bytes_per_spmv": (t2.get("hbm_bytes_per_spmv") or t2.get("hbm_bytes_per_launch")) if fresh2 else None,
                                      "hbm_bytes_per_launch_of_the_dominant_kernel": t2.get("hbm_bytes_per_launch") if fresh2 else None, "kernel": t2.get("kernel"), "measured": t2.get("measured"),
                                      "file": os.path.relpath(tj2, ROOT), "live": False, "plan_fingerprint_matches": fresh2}
                else:
                    rec["traffic"] = None
                rec["seconds"] = round(time.time() - t_wl, 1)
                out["other_workloads"][key] = rec
                api.Tile_destroy(tm2)
                del m2, n2, rp2, ci2, v2, x2, ref2, xd2
            except Exception as e:  # never let an extra break the headline line
                out["other_workloads"][key] = {"error": repr(e)}
        built.clear()
        out["other_workloads_seconds"] = round(time.time() - t_extras, 1)
        # ---- the population (round 5): the committed sweep of 26 generated structures >= 10 M nnz in the class mix of the reference's >= 10 M-nnz matrices
        # (scripts/population_sweep.py -> profiles/r06_population.json: its summary is quoted here), and six of them measured LIVE in this run with the same routine
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import population_sweep as PS
            pop = {"what": "default plans over a population of generated structures (reference's method is a sweep: src/external/CSR5_cuda/bench0.sh:1-14); frac = B_alg / t / 8 TB/s"}
            pj = os.path.join(ROOT, "profiles", "r06_population.json")
            if os.path.exists(pj):
                pjd = json.load(open(pj))
                pop["committed_sweep"] = dict(pjd["summary"], file=os.path.relpath(pj, ROOT), measured=pjd.get("measured"), live=False)
            t_pop = time.time()
            live = {}
            for key, wl, klass in PS.POPULATION:
                if key in PS.LIVE_SUBSET:
                    try:
                        r = PS.measure(key, wl, klass, f64_, torch, api, G, build_matrix, reps=30)
                        live[key] = {k: r[k] for k in ("class", "rows", "nnz", "ms_per_spmv", "frac", "frac_min_bytes", "plan_over_irreducible_bytes", "check_whole_y_exact", "plan_create_seconds", "timed_choices_ms",
                                                     "tile_create_seconds", "tile_format_histogram[csr,coo,ell,hyb,dns,dnsrow,dnscol]")}
                        live[key]["csr_form"] = r["plan"]["csr_form"]; live[key]["entry_mode"] = r["plan"]["entry_mode"]
                    except Exception as e:
                        live[key] = {"error": repr(e)}
                    torch.cuda.empty_cache()
            pop["live_subset"] = dict(live, seconds=round(time.time() - t_pop, 1))
            # this run's own view of the whole line: the live subset + the large other_workloads + the headline
            fr = [v["frac"] for v in live.values() if "frac" in v] + [v["default_plan"]["frac_of_8TBps"] for v in out["other_workloads"].values() if "default_plan" in v] + [round(achieved / HBM_PEAK_GBPS, 4)]
            fm = [v["frac_min_bytes"] for v in live.values() if "frac_min_bytes" in v] + [v["default_plan"]["frac_min_bytes"] for v in out["other_workloads"].values() if "default_plan" in v] + [roofline["frac_min_bytes"]]
            pop["this_run"] = {"count": len(fr), "frac_median": round(float(np.median(fr)), 4), "frac_min": min(fr), "share_frac_ge_0.70": round(sum(f >= 0.70 for f in fr) / len(fr), 3),
                               "share_frac_min_bytes_ge_0.60": round(sum(f >= 0.60 for f in fm) / len(fm), 3),
                               "note": "HBM-bound workloads measured live in this run: headline + large other_workloads + the live subset of the population"}
            out["population"] = pop
        except Exception as e:
            out["population"] = {"error": repr(e)}
        # ---- permuted-numbering plans (round 6; include/tilespmv.h): the window-shuffled meshes in their natural numbering and after reverse Cuthill-McKee — kernel only, and
        # amortised over K products between the two vector permutations (a solver permutes x once at entry and y once at exit: tilespmv_amd/halo.py HaloSpMV(reorder=True) / cg)
        try:
            import reorder_bench as RB
            out["reorder"] = {"what": "default plan of A against the default plan of P A P^T (P: reverse Cuthill-McKee on the symmetrised pattern, host); frac = B_alg / t / 8 TB/s; "
                                      "amortised_over_K_products adds (permute x + un-permute y) / K", "workloads": {}}
            t_ro = time.time()
            for wl in RB.WORKLOADS:
                try:
                    out["reorder"]["workloads"][wl] = RB.measure(wl, torch, api, G, build_matrix, reps=30)
                except Exception as e:
                    out["reorder"]["workloads"][wl] = {"error": repr(e)}
                torch.cuda.empty_cache()
            out["reorder"]["seconds"] = round(time.time() - t_ro, 1)
        except Exception as e:
            out["reorder"] = {"error": repr(e)}
        ph_extras.__exit__(None, None, None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with phase("cpu baseline", 600):
            out["cpu_baseline"] = cpu_baseline(rows, n, rowptr, colidx, vals, x, dtype)
    elif rank == 0:
        out["cpu_baseline"] = None
    out["phase_seconds"] = dict(phase.log)
    if rank == 0:
        full = write_full(out, args.full_json)
        print(compact_line(out, full), flush=True)
    sh.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
