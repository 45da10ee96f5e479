"""Preprocessing on the device (device-side half; reference src/csr2tile.h:629-1020).

``Tile_create_device`` (hip_tile_create.hip: keys -> one stable radix sort -> run-length encoding -> per-tile selection and packing) must hand back the SAME
Tile_matrix as the host ``Tile_create`` (which tests/test_host.py pins byte for byte against the compiled reference): every scalar and every member array
is compared here, on the small all-format / ragged / empty-row cases, on the random ingredients of the fuzz generator (unsorted columns included: the extracted
matrix then needs the reference's pivot sort), on odd sizes (partial last tile-row / tile-column), fp64 and fp32."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from gpu_fuzz import random_matrix  # noqa: E402
from tilespmv_amd import api, generators as G  # noqa: E402
from tilespmv_amd.tile_matrix import to_dict  # noqa: E402

pytestmark = pytest.mark.gpu


def same_tile_matrix(rows, cols, rp, ci, dtype, cdna4=False, real=False, hyb=False):
    nnz = int(rp[rows])
    v = G.real_values(nnz, dtype) if real else G.compat_values(nnz, dtype)
    host = api.Tile_create(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4, hyb=hyb)
    dev = api.Tile_create_device(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4, hyb=hyb)
    try:
        h, d = to_dict(host, rows), to_dict(dev, rows)
        bad = []
        for k in h:
            if isinstance(h[k], np.ndarray):
                if h[k].shape != d[k].shape or h[k].tobytes() != d[k].tobytes():
                    first = int(np.flatnonzero(h[k] != d[k])[0]) if h[k].shape == d[k].shape else -1
                    bad.append("%s (first difference at %d of %d)" % (k, first, h[k].size))
            elif h[k] != d[k]:
                bad.append("%s: host %d device %d" % (k, h[k], d[k]))
        return bad
    finally:
        api.Tile_destroy(host); api.Tile_destroy(dev)


@pytest.mark.parametrize("hyb", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", sorted(cases.SMALL) + sorted(cases.MEDIUM))
def test_device_tile_create_equals_host(name, dtype, hyb):
    """hyb = True: the reference's dormant HYB rule switched on (TILESPMV_CREATE_HYB; width search src/csr2tile.h:279-306, pack :505-548, index bytes :984-1008) — since round 6
    built on the device too, byte for byte the host's Tile_matrix (hybsize / hybellsize / hybcoosize, Blockhyb_Val, hybIdx, the remainders in deferredcoo_*)."""
    rows, cols, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
    assert same_tile_matrix(rows, cols, rp, ci, dtype, hyb=hyb) == []


def test_device_tile_create_builds_hyb_tiles_where_the_rule_selects_them():
    """The HYB cases are not vacuous: the all-formats matrix and the circuit-like stand-in of config 2 really get HYB tiles from both builders."""
    for rows, cols, rp, ci in (cases.SMALL["allfmt"](), G.retarget_nnz(*G.circuit_like(20000, seed=1), target_nnz=112000, seed=1)):
        rows = cases.truncated_rows(rows); nnz = int(rp[rows])
        v = G.compat_values(nnz, np.float64)
        dev = api.Tile_create_device(rows, cols, nnz, rp, ci, v, hyb=True)
        d = to_dict(dev, rows)
        assert int(np.count_nonzero(d["Format"] == 3)) > 0 and d["hybsize"] > 0 and d["hybsize"] == d["hybellsize"] + d["hybcoosize"]
        api.Tile_destroy(dev)
        assert same_tile_matrix(rows, cols, rp, ci, np.float64, hyb=True) == []


def test_device_tile_create_random_ingredients():
    """60 matrices from the fuzz generator's ingredients (every format, long rows, empty tile-rows, odd column counts; half of them with unsorted columns)."""
    for seed in range(60):
        rows, cols, rp, ci = random_matrix(1000 + seed)
        bad = same_tile_matrix(rows, cols, rp, ci, np.float64 if seed % 3 else np.float32, cdna4=seed % 5 == 0, real=seed % 2 == 0, hyb=seed % 4 == 1)
        assert bad == [], "seed %d: %s" % (1000 + seed, bad)


def test_device_tile_create_odd_shapes_and_empty():
    """Partial last tile-row and tile-column, a matrix without nonzeros, a single nonzero."""
    rng = np.random.default_rng(5)
    for rows, cols in [(1, 1), (17, 33), (100, 7), (1000, 999), (31, 5000)]:
        k = min(rows * cols // 3 + 1, 4000)
        key = np.unique(rng.integers(0, rows * cols, k))
        r, c, rp, ci = G.from_coo(rows, cols, key // cols, key % cols)
        assert same_tile_matrix(r, c, rp, ci, np.float64) == [], (rows, cols)
    rp = np.zeros(49, dtype=np.int32); ci = np.zeros(0, dtype=np.int32)
    assert same_tile_matrix(48, 48, rp, ci, np.float64) == []
    r, c, rp, ci = G.from_coo(48, 48, [47], [47])
    assert same_tile_matrix(r, c, rp, ci, np.float32) == []


def test_device_tile_create_large_classes():
    """One matrix of each large class at a size where tile-rows, sorts and scans run many workgroups: stencil, FEM (CSR tiles), power-law (hub rows), KKT."""
    for rows, cols, rp, ci in [G.laplacian5pt(700), G.fem_hex(24, 24, 24, 3), G.powerlaw(400000), G.kkt_like(40), G.rmat(17, 8, 3)]:
        assert same_tile_matrix(rows, cols, rp, ci, np.float64) == []


# ---- the plan built on the device (tilespmv_plan_create_from_csr) against the plan built from the host Tile_matrix

KNOB_SETS = [dict(), dict(deterministic=1), dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2), dict(csr_split=1), dict(csr_split=2), dict(csr_split=2, entry_mode=2),
             dict(dense_mode=1), dict(dense_mode=2), dict(strip_cost=64, split_above=200), dict(csr_split=2, strip_cost=64, split_above=128, dense_mode=1), dict(x_window=2), dict(desc_dict=0),
             dict(desc_dict=1), dict(x_panel_kb=1, x_panel_merge=1, entry_mode=2), dict(wg_strips=32, entry_mode=2), dict(csr_split=3), dict(csr_split=3, entry_mode=2),
             dict(csr_split=3, strip_cost=64, split_above=128, dense_mode=1), dict(csr_split=2, desc_dict=0), dict(csr_split=2, desc_dict=2)]
FACTS = ["device_bytes", "stream_bytes", "nnz", "rows", "tiles", "coo_mode", "dense_mode", "kernel", "num_tasks", "num_split_rows", "entry_mode", "entry_ordered", "strip_cost", "wg_strips", "brick_order",
         "desc_bytes", "nt_stream", "x_panels", "scattered_entries", "csr_form"]


def _spmv(torch, plan, rows, x):
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full((rows + 16,), 12345.0, dtype=xd.dtype, device="cuda")
    plan.spmv(xd.data_ptr(), yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert (y[rows:] == 12345.0).all()
    return y[:rows]


def same_plan(torch, rows, cols, rp, ci, dtype, knobs, cdna4=False, shard=None, hyb=False):
    """Host-built and device-built plan of one matrix and one option set: the same streams (per-stream digests read back from the device), the same facts, the same bits of y."""
    nnz = int(rp[rows])
    v, x = G.real_values(nnz, dtype), G.real_x(cols, nnz, dtype)
    kw = dict(knobs)
    kw.setdefault("placement_tries", 1)
    if shard:
        kw["tilerow_begin"], kw["tilerow_end"] = shard
    tm = api.Tile_create(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4, hyb=hyb)
    host = api.Plan(tm, rows, cols, nnz, **kw)
    dev = api.Plan.from_csr(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4, hyb=hyb, **kw)
    try:
        hi, di = host.info(), dev.info()
        assert di["device_build"] == 1 and hi["device_build"] == 0
        bad = [(k, hi[k], di[k]) for k in FACTS if hi[k] != di[k]]
        assert bad == [], bad
        hs, ds = host.stream_digests(), dev.stream_digests()
        assert sorted(hs) == sorted(ds), (sorted(hs), sorted(ds))
        assert [k for k in hs if hs[k] != ds[k]] == [], {k: (hs[k], ds[k]) for k in hs if hs[k] != ds[k]}
        r0, r1 = (shard[0] * 16, min(rows, shard[1] * 16)) if shard else (0, rows)
        yh, yd = _spmv(torch, host, rows, x), _spmv(torch, dev, rows, x)
        if hi["entry_ordered"]:
            assert np.array_equal(yh[r0:r1], yd[r0:r1])
        else:
            assert np.allclose(yh[r0:r1], yd[r0:r1], rtol=1e-4 if dtype == np.float32 else 1e-11, atol=1e-4 if dtype == np.float32 else 1e-11)
    finally:
        host.close(); dev.close(); api.Tile_destroy(tm)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.gpu
def test_device_built_plan_hub_tile_rows(torch_cuda):
    """Tile-rows that pool tens of thousands of nonzeros from thousands of tiny tiles (R-MAT / web-graph hubs): the device builder fills the pool by one thread per tile at scanned offsets and
    walks the windows with one wavefront per tile-row, 64 candidates per step (windows that straddle steps, chains of one-nonzero windows, full 16-nonzero windows), and packs the entry
    lists one 64-record chunk per step — all of it must give the host builder's streams bit for bit."""
    rng = np.random.default_rng(77)
    n = 60000
    rows_of, cols_of = [], []
    for r in (0, 1, 5, 15, 16, 40, 333, 334):            # hub rows: three in the first tile-row, one alone, two sharing tile-row 20
        k = int(rng.integers(3000, 30000))
        c = np.unique(np.concatenate([rng.integers(0, n, k), np.arange(2000, 2000 + 700)]))   # scattered columns + a dense run (full windows)
        rows_of.append(np.full(len(c), r)); cols_of.append(c)
    k = 4 * n
    rows_of.append(rng.integers(0, n, k)); cols_of.append(rng.integers(0, n, k))                # the rest: 4 per row
    rows_of.append(np.arange(n)); cols_of.append(np.arange(n))
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(sum(len(q) for q in rows_of)), (np.concatenate(rows_of), np.concatenate(cols_of))), shape=(n, n))
    A.sum_duplicates(); A.sort_indices()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    for knobs in (dict(csr_split=2), dict(csr_split=3), dict(csr_split=2, entry_mode=2), dict(csr_split=3, entry_mode=1), dict(entry_mode=2, x_panel_kb=1, x_panel_merge=1), dict()):
        same_plan(torch_cuda, n, n, rp, ci, np.float64, knobs)
    same_plan(torch_cuda, n, n, rp, ci, np.float32, dict(csr_split=3, entry_mode=2))
    same_plan(torch_cuda, n, n, rp, ci, np.float64, dict(csr_split=2), shard=(0, 21))


@pytest.mark.parametrize("name", sorted(cases.SMALL) + sorted(cases.MEDIUM))
def test_device_built_plan_equals_host_built_plan(torch_cuda, name):
    rows, cols, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
    rows = cases.truncated_rows(rows)
    for i, knobs in enumerate(KNOB_SETS):
        same_plan(torch_cuda, rows, cols, rp, ci, np.float64 if i % 2 == 0 else np.float32, knobs)


def test_device_built_plan_classes_and_shards(torch_cuda):
    """FEM (pooled units), 3-D stencil and KKT (brick order, dictionary), power-law (workgroup entry lists, split rows), band (dense tiles on the matrix cores), R-MAT;
    whole matrix and a shard of tile-rows, fp64 and fp32."""
    mats = {"fem3": G.fem_hex(14, 14, 14, 3), "fem6s": G.fem_hex(10, 10, 10, 6, shuffle=16), "fem3s64": G.fem_hex(16, 16, 16, 3, shuffle=64), "circuit": G.circuit_like(120000), "lap3d": G.laplacian7pt(48), "kkt24": G.kkt_like(24), "powerlaw": G.powerlaw(300000),
            "band40": G.band(60000, 40), "rmat16": G.rmat(16, 8, 3), "lap2d": G.laplacian5pt(500)}
    for name, (rows, cols, rp, ci) in mats.items():
        rows = cases.truncated_rows(rows)
        tilem = rows // 16
        for dtype in (np.float64, np.float32):
            same_plan(torch_cuda, rows, cols, rp, ci, dtype, dict())
            same_plan(torch_cuda, rows, cols, rp, ci, dtype, dict(deterministic=1), shard=(tilem // 3, 2 * tilem // 3))


def test_device_built_plan_random_ingredients(torch_cuda):
    for seed in range(40):
        rows, cols, rp, ci = random_matrix(2000 + seed)
        knobs = KNOB_SETS[seed % len(KNOB_SETS)]
        same_plan(torch_cuda, rows, cols, rp, ci, np.float64 if seed % 2 else np.float32, knobs, cdna4=seed % 7 == 0, hyb=seed % 3 == 1)


def test_device_built_plan_with_hyb_tiles(torch_cuda):
    """Config 2's requirement (all seven formats) through tilespmv_plan_create_from_csr: with TILESPMV_CREATE_HYB the device-built plan has the host-built plan's streams, facts and y —
    split, pooled and wide forms, per-strip / per-wavefront / per-workgroup entry lists (HYB remainders go where COO entries go)."""
    from oracle.oracle import CpuImpl
    mats = [cases.SMALL["allfmt"](), G.retarget_nnz(*G.circuit_like(60000, seed=1), target_nnz=336000, seed=1)]
    for rows, cols, rp, ci in mats:
        rows = cases.truncated_rows(rows)
        for i, knobs in enumerate([dict(), dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2), dict(csr_split=1), dict(csr_split=2), dict(csr_split=3), dict(deterministic=1, strip_cost=64, split_above=200)]):
            same_plan(torch_cuda, rows, cols, rp, ci, np.float64 if i % 2 == 0 else np.float32, knobs, hyb=True)
    # ... and the result is the oracle's (HYB oracle = the reference's headers with the dormant branch re-enabled, oracle/Makefile), bit for bit on integer data
    rows, cols, rp, ci = mats[1]
    rows = cases.truncated_rows(rows); nnz = int(rp[rows])
    for dtype in (np.float64, np.float32):
        O = CpuImpl("oracle", dtype)
        v, x = G.compat_values(nnz, dtype), G.compat_x(cols, dtype)
        want = O.spmv(O.tile_create(rows, cols, nnz, rp, ci, v, hyb=True), rows, cols, nnz, rp, ci, v, x)["y"]
        p = api.Plan.from_csr(rows, cols, nnz, rp, ci, v, dtype=dtype, hyb=True)
        assert np.array_equal(_spmv(torch_cuda, p, rows, x), want)
        p.close()


def test_device_built_plan_matches_the_oracle(torch_cuda):
    """y of device-built plans against the oracle's CSR golden, bit-exact on the reference's small-integer data (the same bar as every other plan)."""
    from oracle.oracle import CpuImpl
    for name in ["allfmt", "allfmt_pad5", "circuit8k", "powerlaw20k", "one_long_row", "kkt12"]:
        rows, cols, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
        rows = cases.truncated_rows(rows)
        nnz = int(rp[rows])
        for dtype in (np.float64, np.float32):
            v, x = cases.values_for(name, nnz, cols, dtype)
            O = CpuImpl("oracle", dtype)
            to = O.tile_create(rows, cols, nnz, rp, ci, v)
            want = O.spmv(to, rows, cols, nnz, rp, ci, v, x)["y"]
            for knobs in (dict(), dict(csr_split=2), dict(entry_mode=2, dense_mode=1)):
                plan = api.Plan.from_csr(rows, cols, nnz, rp, ci, v, dtype=dtype, **knobs)
                y = _spmv(torch_cuda, plan, rows, x)
                plan.close()
                assert np.array_equal(y, want), (name, dtype, knobs, int(np.count_nonzero(y != want)))


def test_options_without_a_device_path_are_refused(torch_cuda):
    rows, cols, rp, ci = cases.SMALL["lap64"]()
    v = G.compat_values(len(ci), np.float64)
    for knobs in (dict(csr_split=0), dict(kernel=api.KERNEL_DIRECT), dict(coo_mode=api.COO_FALLBACK)):
        with pytest.raises(NotImplementedError):
            api.Plan.from_csr(rows, cols, len(ci), rp, ci, v, **knobs)


def test_measured_selection_on_the_device_path(torch_cuda):
    """autotune = 1 through tilespmv_plan_create_from_csr: the candidates are built from the one device-resident tiled matrix and timed; whatever wins, y is the oracle's (integer data: exact),
    the plan says it was built on the device, and the log has the candidates."""
    import json, tempfile
    from oracle.oracle import CpuImpl
    O = CpuImpl("oracle", np.float64)
    for name, gen in (("powerlaw200k", cases.MEDIUM["powerlaw200k"]), ("fem", lambda: G.fem_hex(12, 12, 12, 3)), ("allfmt", cases.SMALL["allfmt"])):
        rows, cols, rp, ci = gen()
        rows = cases.truncated_rows(rows); nnz = int(rp[rows])
        v, x = cases.values_for(name, len(ci), cols, np.float64)
        want = O.spmv(O.tile_create(rows, cols, nnz, rp, ci, v), rows, cols, nnz, rp, ci, v, x)["y"]
        log = tempfile.mktemp(suffix=".jsonl")
        os.environ["TILESPMV_AUTOTUNE_LOG"] = log
        try:
            p = api.Plan.from_csr(rows, cols, nnz, rp, ci, v, autotune=True)
        finally:
            os.environ.pop("TILESPMV_AUTOTUNE_LOG", None)
        assert p.info()["device_build"] == 1
        assert np.array_equal(_spmv(torch_cuda, p, rows, x), want), name
        p.close()
        rec = json.loads(open(log).read().strip().splitlines()[-1])
        assert len([c for c in rec["candidates"] if c.get("label") != "confirm"]) >= 3 and "choice" in rec, rec
        os.remove(log)


def test_cli_with_the_plan_built_on_the_device(torch_cuda, tmp_path):
    """`TILESPMV_DEVICE_BUILD=1 ./test -d 0 test.mtx`: call_tilespmv_hip builds its plan from the CSR arguments on the device; same lines, Check... PASS."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for exe in ("test_f64", "test_f32"):
        env = dict(os.environ, TILESPMV_WARMUP="2", TILESPMV_BENCH_REPEAT="5", TILESPMV_DEVICE_BUILD="1", TILESPMV_PLAN_VERBOSE="1")
        r = subprocess.run([os.path.join(root, "tilespmv_amd", "bin", exe), "-d", "0", os.path.join(root, "tests", "golden", "test.mtx")], cwd=tmp_path, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert "Run CPU TileSpMV, errcount = 0" in r.stdout and "Check... PASS!" in r.stdout
        assert "plan from CSR: device Tile_create" in r.stderr   # the device path really ran
    # the CLI's own switch: Tile_create_device + device-built plan, the reference's lines in order, one added line
    env2 = dict(os.environ, TILESPMV_WARMUP="2", TILESPMV_BENCH_REPEAT="5", TILESPMV_PLAN_VERBOSE="1")
    r = subprocess.run([os.path.join(root, "tilespmv_amd", "bin", "test_f64"), "-d", "0", os.path.join(root, "tests", "golden", "test.mtx"), "--device-build"], cwd=tmp_path, env=env2,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    pos = -1
    for w in ["The number of tile =", "device build: Tile_matrix created on the device", "Run CPU TileSpMV, errcount = 0", "CUDA SpMV runtime", "Check... PASS!"]:
        nxt = r.stdout.find(w, pos + 1)
        assert nxt > pos, (w, r.stdout)
        pos = nxt
    assert "plan from CSR: device Tile_create" in r.stderr
    # the multi-device driver: every shard tiles ITS row block of the CSR arguments on its device (eight shards on the one device there is)
    for extra in (["--combine=none"], ["--combine=allgather"]):
        r = subprocess.run([os.path.join(root, "tilespmv_amd", "bin", "test_f64"), "-d", "0,0,0,0,0,0,0,0", os.path.join(root, "tests", "golden", "test.mtx")] + extra, cwd=tmp_path, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, (extra, r.stdout, r.stderr)
        assert "HIP SpMV on 8 device(s)" in r.stdout and "Check... PASS!" in r.stdout
        assert r.stderr.count("plan from CSR: device Tile_create") == 8


def test_row_block_of_a_larger_csr(torch_cuda):
    """tilespmv_plan_create_from_csr on a slice of a matrix's row pointer (pointers not rebased, as the multi-device driver passes them): the plan of that row block."""
    import ctypes as C
    from tilespmv_amd import _lib
    rows, cols, rp, ci = G.fem_hex(12, 12, 12, 3)
    lib = _lib.load(np.float64)
    v = G.real_values(len(ci), np.float64); x = G.real_x(cols, len(ci), np.float64)
    rp32, ci32 = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32)
    r0, r1 = 16 * 40, 16 * 200
    h = C.c_void_p()
    opts = _lib.PlanOptions(deterministic=1)
    rc = lib.tilespmv_plan_create_from_csr(C.byref(h), r1 - r0, cols, int(rp32[r1] - rp32[r0]), rp32[r0:].ctypes.data_as(_lib._I), ci32.ctypes.data_as(_lib._I), v.ctypes.data_as(C.POINTER(C.c_double)),
                                           api.CREATE_QUIET, C.byref(opts))
    assert rc == 0
    xd = torch_cuda.from_numpy(x).cuda(); yd = torch_cuda.zeros(r1 - r0 + 16, dtype=xd.dtype, device="cuda")
    assert lib.tilespmv_plan_spmv(h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), None) == 0
    torch_cuda.cuda.synchronize()
    lib.tilespmv_plan_destroy(h)
    import scipy.sparse as sp
    want = sp.csr_matrix((v, ci, rp), shape=(rows, cols))[r0:r1] @ x
    assert np.allclose(yd.cpu().numpy()[:r1 - r0], want, rtol=1e-11, atol=1e-11)


def test_device_built_plan_degenerate_inputs(torch_cuda):
    """Empty matrix, one nonzero, fewer than 16 rows, a partial last tile-row and tile-column, an empty first / last tile-row: the device-built plan against scipy."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    cases_ = [(48, 48, [], []), (48, 48, [47], [47]), (16, 16, [0], [0]), (160, 160, [40, 41, 150], [5, 100, 159]), (1000, 999, None, None), (33, 700, None, None), (17, 17, None, None)]
    for rows, cols, ri, cj in cases_:
        if ri is None:
            k = min(rows * cols // 4 + 1, 5000)
            key = np.unique(rng.integers(0, rows * cols, k)); ri, cj = key // cols, key % cols
        r, c, rp, ci = G.from_coo(rows, cols, np.asarray(ri, dtype=np.int64), np.asarray(cj, dtype=np.int64))
        rows16 = (r // 16) * 16   # (the driver rule: whole tile-rows)
        if rows16 == 0:
            continue
        nnz = int(rp[rows16])
        for dtype in (np.float64, np.float32):
            v, x = G.compat_values(max(nnz, 1), dtype)[:nnz], G.compat_x(c, dtype)
            for knobs in (dict(), dict(csr_split=2), dict(csr_split=3), dict(entry_mode=2)):
                plan = api.Plan.from_csr(rows16, c, nnz, rp[:rows16 + 1], ci[:nnz], v, dtype=dtype, **knobs)
                y = _spmv(torch_cuda, plan, rows16, x)
                plan.close()
                want = sp.csr_matrix((v.astype(np.float64), ci[:nnz], rp[:rows16 + 1]), shape=(rows16, c)) @ x.astype(np.float64)
                assert np.array_equal(y.astype(np.float64), want), (rows, cols, dtype, knobs)


def test_plan_from_a_device_resident_csr(torch_cuda):
    """tilespmv_plan_create_from_device_csr: the CSR arrays are torch tensors on the GPU (what a torch CSR tensor holds): the same plan as from the host arrays, stream for stream; timing beside it."""
    import time
    for gen, dtype in ((G.fem_hex(20, 20, 20, 3), np.float64), (G.powerlaw(300000), np.float32), (G.laplacian5pt(600), np.float64)):
        rows, cols, rp, ci = gen
        rows = cases.truncated_rows(rows); nnz = int(rp[rows])
        v, x = G.real_values(nnz, dtype), G.real_x(cols, nnz, dtype)
        host = api.Plan.from_csr(rows, cols, nnz, rp, ci, v, dtype=dtype, deterministic=1)
        d_rp = torch_cuda.from_numpy(np.ascontiguousarray(rp[:rows + 1], dtype=np.int32)).cuda(); d_ci = torch_cuda.from_numpy(np.ascontiguousarray(ci[:nnz], dtype=np.int32)).cuda()
        d_v = torch_cuda.from_numpy(np.ascontiguousarray(v[:nnz])).cuda()
        torch_cuda.cuda.synchronize()
        dev = api.Plan.from_device_csr(rows, cols, nnz, d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), dtype, deterministic=1)
        try:
            a, b = host.stream_digests(), dev.stream_digests()
            assert sorted(a) == sorted(b) and all(a[k] == b[k] for k in a)
            assert np.array_equal(_spmv(torch_cuda, host, rows, x), _spmv(torch_cuda, dev, rows, x))
            assert d_rp.cpu().numpy().tolist()[:3] == rp[:3].tolist()   # the caller's arrays are untouched
        finally:
            host.close(); dev.close()
