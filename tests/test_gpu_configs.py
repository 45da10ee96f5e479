"""GPU parity at the full size of the BASELINE configs (run with -m gpu on an MI355X).

Config 2 (scircuit) and config 3 (webbase-1M) stand-ins are sized like the SuiteSparse matrices they stand for
(reference src/external/CSR5_cuda/2757-matrix.csv:544 170,998^2 / 958,936 nnz; :2379 1,000,005^2 / 3,105,536 nnz); the
oracle finishes them in seconds, so the WHOLE y is compared bit for bit, in every execution mode.  Config 5
(nlpkkt160, :1903 8,345,600^2 / 229,518,112 nnz, fp32) is checked through properties.  Any real Matrix Market file
found under $TILESPMV_MATRIX_DIR goes through the CLI and through the plan API against the oracle.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def _gpu_y(torch, tp, rowA, n, nnz, x, **kw):
    from tilespmv_amd import api
    plan = api.Plan(tp, rowA, n, nnz, **kw)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full((rowA + 16,), 12345.0, dtype=xd.dtype, device="cuda")
    plan.spmv(xd.data_ptr(), yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert (y[rowA:] == 12345.0).all(), "wrote past the end of y"
    info = plan.info()
    plan.close()
    return y[:rowA], info


def _all_modes(torch, name, dtype, hyb):
    """Whole-y bit-exact comparison of every kernel generation / COO mode / dense mode with the oracle."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    from tilespmv_amd.tile_matrix import field_array
    m, n, rp, ci, _ = _bench().build_matrix(name)
    rowA = (m // 16) * 16                                   # the driver rule of the reference (src/main.cu:71)
    nnz = len(ci)
    vals, x = G.compat_values(nnz, dtype), G.compat_x(n, dtype)
    O = CpuImpl("oracle", dtype)
    to = O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb)
    want = O.spmv(to, rowA, n, nnz, rp, ci, vals, x)
    assert want["errcount"] == 0
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb)
    hist = np.bincount(field_array(tp, "Format", tp.tilenum), minlength=7)
    seen = set()
    for kernel in (api.KERNEL_STREAM, api.KERNEL_DIRECT):
        for coo in (api.COO_IN_TILE, api.COO_FALLBACK, api.COO_AUTO):
            for dns in (api.DENSE_MFMA, api.DENSE_VALU):
                y, info = _gpu_y(torch, tp, rowA, n, nnz, x, coo_mode=coo, dense_mode=dns, kernel=kernel)
                assert np.array_equal(y, want["y"]), (name, kernel, coo, dns, int(np.count_nonzero(y != want["y"])))
                seen.add((info["kernel"], info["coo_mode"], info["dense_mode"]))
    assert len(seen) == 8                                    # AUTO resolves to one of the two explicit COO modes
    for env in ({"TILESPMV_WAVE_COO": "0"}, {"TILESPMV_WAVE_COO": "1"}, {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "1"},
                {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "0"}):
        os.environ.update(env)
        try:
            y, _ = _gpu_y(torch, tp, rowA, n, nnz, x)
        finally:
            for k in env:
                os.environ.pop(k)
        assert np.array_equal(y, want["y"]), (name, env)
    api.Tile_destroy(tp)
    return m, n, nnz, hist, tp


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config2_scircuit_full_size_every_mode(torch_cuda, dtype):
    """170,998 rows / 958,936 nnz, HYB rule on: all seven tile formats occur, whole y == oracle in every mode."""
    m, n, nnz, hist, _ = _all_modes(torch_cuda, "scircuit", dtype, hyb=True)
    assert (m, n, nnz) == (170998, 170998, 958936)
    assert (hist > 0).all(), hist.tolist()                   # csr, coo, ell, hyb, dns, dnsrow, dnscol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_config3_webbase_full_size_every_mode(torch_cuda, dtype):
    """1,000,005 rows / 3,105,536 nnz, last tile column has 5 columns; the very-sparse path in both COO modes."""
    m, n, nnz, hist, _ = _all_modes(torch_cuda, "webbase", dtype, hyb=False)
    assert (m, n, nnz) == (1000005, 1000005, 3105536) and n % 16 == 5
    assert hist[1] > 0.9 * hist.sum()                        # >= 90 % COO tiles


def test_config5_nlpkkt160_full_size_f32(torch_cuda):
    """8,345,600 rows / 229,518,112 nnz in fp32: whole y against the CSR golden (integer data: every partial sum is
    exact in fp32), linearity, idempotent relaunch.
    Checked against the CSR golden (y_golden = CSR product of the same data) — the reference's own criterion for its GPU result (src/main.cu:101-110 builds it,
    :186-197 compares) — not against tilespmv_cpu: the oracle's serial tile loop does not finish in test time at this size; the tile path itself is pinned on the
    small and medium cases (tests/test_gpu_parity.py against oracle/, tests/test_host.py against oracle/_ref)."""
    import torch
    from tilespmv_amd import api, generators as G
    m, n, rp, ci, _ = _bench().build_matrix("nlpkkt160")
    nnz = len(ci)
    assert (m, n, nnz) == (G.NLPKKT160_ROWS, G.NLPKKT160_ROWS, G.NLPKKT160_NNZ) and m % 16 == 0
    vals = G.compat_values(nnz, np.float32)
    tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=np.float32)
    plan = api.Plan(tp, m, n, nnz)
    api.Tile_destroy(tp)
    rng = np.random.default_rng(1)
    x1 = rng.integers(0, 4, n).astype(np.float32); x2 = rng.integers(0, 4, n).astype(np.float32)
    ys = []
    for x in (x1, x2, x1 + x2):
        xd = torch.from_numpy(x).cuda(); yd = torch.zeros(m + 16, dtype=torch.float32, device="cuda")
        plan.spmv(xd.data_ptr(), yd.data_ptr()); plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
        ys.append(yd.cpu().numpy()[:m])
    assert (np.diff(rp) > 0).all()
    seg = np.add.reduceat(vals.astype(np.float64) * x1[ci].astype(np.float64), rp[:-1])
    assert np.array_equal(ys[0].astype(np.float64), seg)
    assert np.array_equal(ys[0] + ys[1], ys[2])
    plan.close()


@pytest.mark.parametrize("workload", ["bandrand4x3_2000000", "uniform8_2000000", "powerlaw8000000"])
def test_irregular_class_default_plans_full_size(torch_cuda, workload):
    """The north star's "synthetic banded / power-law" side and the scattered class VERDICT round 3 named, at the size the bench line quotes them (uniform random at half of it), DEFAULT plan:
    the two scattered matrices record column panels by rule (x >= 12 MB, entry-dominated) and decide by timing between the plain launch, panelled launches and column slices pinned to XCDs; whatever is chosen, the whole y equals the
    CSR golden bit for bit (integer data), twice in a row, and y = A (x1 + x2) = A x1 + A x2."""
    import torch
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    m, n, rp, ci, _ = _bench().build_matrix(workload)
    rowA = (m // 16) * 16; nnz = int(rp[rowA])
    vals = G.compat_values(len(ci))
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals)
    plan = api.Plan(tp, rowA, n, nnz)
    api.Tile_destroy(tp)
    info = plan.info()
    assert info["entry_mode"] == 2 and info["x_panels"] >= 1
    assert info["x_panels"] == (info["x_slice_passes"] if info["x_slice_passes"] > 0 else 1 if info["x_panel_merge"] == 0 else info["x_panels"])
    assert info["x_slice_passes"] == 0 or (info["x_panel_merge"] == 0 and info["entry_ordered"] == 0)    # column slices on XCDs: sums meet in any order, and the plan says so
    if workload.startswith("powerlaw"):
        assert info["x_panels"] == 1          # most entries sit near the diagonal: every pass would re-read y for a handful of entries (the timing drops the panels)
    rng = np.random.default_rng(4)
    x1 = rng.integers(0, 4, n).astype(np.float64); x2 = rng.integers(0, 4, n).astype(np.float64)
    O = CpuImpl("oracle")
    ys = []
    for x in (x1, x2, x1 + x2):
        xd = torch.from_numpy(x).cuda(); yd = torch.full((rowA + 16,), -3.0, dtype=torch.float64, device="cuda")
        plan.spmv(xd.data_ptr(), yd.data_ptr()); plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
        y = yd.cpu().numpy()
        assert (y[rowA:] == -3.0).all()
        ys.append(y[:rowA])
    assert np.array_equal(ys[0], O.csr_spmv(rowA, rp, ci, vals, x1)), workload
    assert np.array_equal(ys[0] + ys[1], ys[2])
    plan.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("workload", ["fem3_68", "fem6_46", "fem3s64_68"])
def test_fem_class_default_plans_full_size(torch_cuda, workload, dtype):
    """Round 5 (Missing 1): the FEM / block-structured class — > 90 % of the nonzeros in CSR-format tiles — at the size the bench line quotes it (74-91 M nnz), DEFAULT plan,
    fp64 and fp32.  The byte model must choose the pooled units (csr_form 2: CSR tiles, COO tiles and HYB remainders pooled per tile-row; fem6 also runs its 99 k dense tiles on the matrix
    cores; the window-shuffled mesh gets the wide windows of csr_form 3), the plan's streams must come to at most 0.82 x the CSR-model bytes in fp64 (VERDICT's mark), the whole y must equal the CSR golden bit for bit — also with reproducible
    sums forced in the workgroup entry mode — and y = A (x1 + x2) = A x1 + A x2."""
    import torch
    from tilespmv_amd import api, generators as G
    m, n, rp, ci, _ = _bench().build_matrix(workload)
    rowA = (m // 16) * 16; nnz = int(rp[rowA])
    vals = G.compat_values(len(ci), dtype)
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rng = np.random.default_rng(4)
    x1 = rng.integers(0, 4, n).astype(dtype); x2 = rng.integers(0, 4, n).astype(dtype)
    seg = lambda x: np.add.reduceat(vals[:nnz].astype(np.float64) * x[ci[:nnz]].astype(np.float64), rp[:rowA])   # every row has entries
    assert (np.diff(rp[:rowA + 1]) > 0).all()
    for kw in (dict(), dict(entry_mode=2, entry_ordered=1)):
        plan = api.Plan(tp, rowA, n, nnz, **kw)
        info = plan.info()
        # natural-order meshes: 16-column pooled units, a few dozen patterns -> 8-byte descriptors + dictionary; the window-shuffled one: wide pooled units (256-column windows, 28-byte descriptors)
        assert (info["csr_form"], info["desc_bytes"]) == ((3, 28) if workload == "fem3s64_68" else (2, 4)), (workload, info)
        if dtype == np.float64 and workload != "fem3s64_68":
            assert info["stream_bytes"] <= 0.82 * api.algorithmic_bytes(nnz, rowA, n, 8), (workload, info["stream_bytes"])
        ys = []
        for x in (x1, x2, x1 + x2):
            xd = torch.from_numpy(x).cuda(); yd = torch.full((rowA + 16,), -3.0, dtype=tdt, device="cuda")
            plan.spmv(xd.data_ptr(), yd.data_ptr()); plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
            y = yd.cpu().numpy()
            assert (y[rowA:] == -3.0).all()
            ys.append(y[:rowA])
        assert np.array_equal(ys[0].astype(np.float64), seg(x1)), (workload, kw)
        assert np.array_equal(ys[0] + ys[1], ys[2])
        plan.close()
    api.Tile_destroy(tp)


def test_wide_band_dense_pieces_with_large_strip_cost(torch_cuda, monkeypatch):
    """A tile-row with more dense tiles than one matrix-core piece may hold (k_dense_mfma broadcasts the column blocks of
    a piece from one 64-lane load) must be cut into pieces whatever the cost knobs say: band with half-bandwidth 640
    (80 dense tiles per tile-row) under TILESPMV_STRIP_COST=800, which used to keep such rows whole."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    m, n, rp, ci = G.band(2048, 640)
    nnz = len(ci)
    vals, x = G.compat_values(nnz), G.compat_x(n)
    want = CpuImpl("oracle").csr_spmv(m, rp, ci, vals, x)
    tp = api.Tile_create(m, n, nnz, rp, ci, vals)
    for cost in ("800", "1600", "400"):
        monkeypatch.setenv("TILESPMV_STRIP_COST", cost)
        for dns in (api.DENSE_MFMA, api.DENSE_VALU):
            y, info = _gpu_y(torch_cuda, tp, m, n, nnz, x, dense_mode=dns)
            assert np.array_equal(y, want), (cost, dns)
            if dns == api.DENSE_MFMA:
                assert info["num_split_rows"] > 0
    monkeypatch.delenv("TILESPMV_STRIP_COST")
    # multi-vector form of the same plan kind
    import torch
    X = np.stack([x, x[::-1].copy()], axis=1).copy()
    plan = api.Plan(tp, m, n, nnz, dense_mode=api.DENSE_MFMA)
    Xd = torch.from_numpy(X).cuda(); Yd = torch.zeros((m + 16, 2), dtype=torch.float64, device="cuda")
    plan.spmm(Xd.data_ptr(), Yd.data_ptr(), 2); torch.cuda.synchronize()
    Y = Yd.cpu().numpy()[:m]
    assert np.array_equal(Y[:, 0], want) and np.array_equal(Y[:, 1], CpuImpl("oracle").csr_spmv(m, rp, ci, vals, X[:, 1].copy()))
    plan.close()
    api.Tile_destroy(tp)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_real_values_bitwise_where_the_order_cannot_matter_and_run_to_run(torch_cuda, dtype):
    """Non-integer data.  (1) A matrix with at most two entries per row: a two-term sum is commutative, so the GPU must
    match the oracle BIT FOR BIT although it adds in another order — values and x really travel unrounded.  (2) On an
    irregular matrix (long rows cut into pieces, LDS scatter-adds) five launches give identical bits: the order of the
    additions is fixed by the plan, not by timing (the reference's atomicAdd, src/tilespmv_cuda.h:784-790, is not)."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    rng = np.random.default_rng(7)
    rows = 20000
    ri = np.concatenate([np.arange(rows), np.arange(rows)])
    cj = np.concatenate([rng.integers(0, rows, rows), rng.integers(0, rows, rows)])
    m, n, rp, ci = G.from_coo(rows, rows, ri, cj)
    assert np.diff(rp).max() <= 2
    nnz = len(ci)
    vals = rng.uniform(-1, 1, nnz).astype(dtype); x = rng.uniform(-1, 1, n).astype(dtype)
    O = CpuImpl("oracle", dtype)
    want = O.spmv(O.tile_create(m, n, nnz, rp, ci, vals), m, n, nnz, rp, ci, vals, x)["y"]
    tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dtype)
    for kw in ({}, {"coo_mode": api.COO_FALLBACK}, {"kernel": api.KERNEL_DIRECT}):
        y, _ = _gpu_y(torch_cuda, tp, m, n, nnz, x, **kw)
        assert np.array_equal(y.view(np.uint8), want.view(np.uint8)), kw
    api.Tile_destroy(tp)
    m, n, rp, ci = G.powerlaw(200000)
    m = (m // 16) * 16
    nnz = len(ci)
    vals = rng.uniform(-1, 1, nnz).astype(dtype); x = rng.uniform(-1, 1, n).astype(dtype)
    tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dtype)
    for env in ({}, {"TILESPMV_WAVE_COO": "0"}, {"TILESPMV_WAVE_COO": "1"}, {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "1"},
                {"TILESPMV_SPLIT_ABOVE": "300", "TILESPMV_STRIP_COST": "48"}):
        os.environ.update(env)
        try:
            runs = [_gpu_y(torch_cuda, tp, m, n, nnz, x) for _ in range(5)]
        finally:
            for k in env:
                os.environ.pop(k)
        assert runs[0][1]["entry_ordered"] == 1, env          # the plan says its sums are reproducible ...
        assert all(np.array_equal(runs[0][0].view(np.uint8), r[0].view(np.uint8)) for r in runs[1:]), env   # ... and they are
    api.Tile_destroy(tp)


@pytest.mark.parametrize("mode", ["0", "1", "2u", "2o"])
def test_entry_modes_real_values_within_tolerance(torch_cuda, mode):
    """Every entry mode on real-valued data: |y - y_ref| <= tol * sum_j |a_ij x_j| (1e-12 fp64 / 1e-5 fp32),
    on an irregular matrix with split rows, in both dtypes."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    env = {"0": {"TILESPMV_WAVE_COO": "0"}, "1": {"TILESPMV_WAVE_COO": "1"}, "2u": {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "0"},
           "2o": {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "1"}}[mode]
    m, n, rp, ci = G.powerlaw(200000)
    m = (m // 16) * 16
    nnz = len(ci)
    rng = np.random.default_rng(11)
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        vals = rng.uniform(-1, 1, nnz).astype(dtype); x = rng.uniform(-1, 1, n).astype(dtype)
        O = CpuImpl("oracle", dtype)
        want = O.spmv(O.tile_create(m, n, nnz, rp, ci, vals), m, n, nnz, rp, ci, vals, x)["y"].astype(np.float64)
        ri = np.repeat(np.arange(m), np.diff(rp[:m + 1]))
        bound = np.zeros(m); np.add.at(bound, ri, np.abs(vals[:rp[m]].astype(np.float64) * x[ci[:rp[m]]].astype(np.float64)))
        tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dtype)
        os.environ.update(env)
        try:
            y, info = _gpu_y(torch_cuda, tp, m, n, nnz, x)
        finally:
            for k in env:
                os.environ.pop(k)
        assert info["entry_mode"] == int(mode[0])
        assert (np.abs(y.astype(np.float64) - want) <= tol * bound + 1e-300).all(), (mode, dtype)
        api.Tile_destroy(tp)


def test_non_finite_x_reaches_only_what_the_header_says(torch_cuda):
    """include/tilespmv.h: x must be finite.  What happens otherwise is pinned here so that it cannot change silently: rows
    that store an entry in the Inf column become non-finite, and so may rows whose tiles hold zero PADDING in that column
    block (ELL slots as in the reference, units, dense tiles); rows whose tile-row has no tile in that column block stay
    exactly as they were."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    m, n, rp, ci = G.laplacian5pt(64)
    nnz = len(ci)
    vals, x = G.compat_values(nnz), G.compat_x(n)
    base = CpuImpl("oracle").csr_spmv(m, rp, ci, vals, x)
    bad = 1000
    x2 = x.copy(); x2[bad] = np.inf
    tp = api.Tile_create(m, n, nnz, rp, ci, vals)
    y, _ = _gpu_y(torch_cuda, tp, m, n, nnz, x2)
    ri = np.repeat(np.arange(m), np.diff(rp))
    direct = np.unique(ri[ci == bad])                               # rows with a stored entry in that column
    assert not np.isfinite(y[direct]).any()
    touched = np.unique(ri[(ci // 16) == bad // 16] // 16)          # tile-rows with a tile in that column block
    clean = np.ones(m, bool)
    for tr in touched:
        clean[16 * tr:16 * tr + 16] = False
    assert np.array_equal(y[clean], base[clean])
    api.Tile_destroy(tp)


def _mtx_files():
    d = os.environ.get("TILESPMV_MATRIX_DIR")
    return sorted(glob.glob(os.path.join(d, "*.mtx"))) if d and os.path.isdir(d) else []


def test_real_matrix_files_through_cli_and_plan(torch_cuda, tmp_path):
    """Every *.mtx under $TILESPMV_MATRIX_DIR (e.g. scircuit.mtx, webbase-1M.mtx): the `test` CLI must PASS, and the plan
    API must match the oracle bit for bit on the reference's driver data.  Skipped when no file is present (there is
    no network on the build machines: SURVEY S7)."""
    files = _mtx_files()
    if not files:
        pytest.skip("no $TILESPMV_MATRIX_DIR/*.mtx present")
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    exe = os.path.join(ROOT, "tilespmv_amd", "bin", "test_f64")
    env = dict(os.environ, TILESPMV_WARMUP="2", TILESPMV_BENCH_REPEAT="5")
    for f in files:
        if os.path.getsize(f) > (2 << 30):
            continue                                          # multi-GB text files: bench.py's job, not the test suite's
        r = subprocess.run([exe, "-d", "0", f], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0 and "Check... PASS!" in r.stdout and "errcount = 0" in r.stdout, (f, r.stdout[-500:], r.stderr[-500:])
        mm = api.mmio_allinone(f)
        assert mm["rc"] == 0
        m, n, rp, ci = mm["m"], mm["n"], mm["rowptr"], mm["colidx"]
        rowA, nnz = (m // 16) * 16, mm["nnz"]
        vals, x = G.compat_values(nnz), G.compat_x(n)
        O = CpuImpl("oracle")
        want = O.spmv(O.tile_create(rowA, n, nnz, rp, ci, vals), rowA, n, nnz, rp, ci, vals, x)["y"]
        tp = api.Tile_create(rowA, n, nnz, rp, ci, vals)
        for kw in ({}, {"coo_mode": api.COO_FALLBACK}, {"kernel": api.KERNEL_DIRECT}):
            y, _ = _gpu_y(torch_cuda, tp, rowA, n, nnz, x, **kw)
            assert np.array_equal(y, want), (f, kw)
        api.Tile_destroy(tp)


def test_matrix_dir_hook_with_a_generated_file(torch_cuda, tmp_path, monkeypatch):
    """The $TILESPMV_MATRIX_DIR hook itself, exercised with a file written here: bench.build_matrix picks the file over
    the stand-in, and the file goes through the same CLI + plan checks as a real SuiteSparse file would."""
    from tilespmv_amd import generators as G
    m, n, rp, ci = G.circuit_like(4000, seed=3)
    G.write_mtx(str(tmp_path / "scircuit.mtx"), m, n, rp, ci)
    monkeypatch.setenv("TILESPMV_MATRIX_DIR", str(tmp_path))
    bm = _bench().build_matrix("scircuit")
    assert bm[4] == "file:scircuit.mtx" and bm[0] == m and len(bm[3]) == len(ci)
    test_real_matrix_files_through_cli_and_plan(torch_cuda, tmp_path)


def test_bench_takes_any_mtx_file(torch_cuda, tmp_path):
    """`bench.py --workload path/to/A.mtx` (a SuiteSparse download, round 6): the file goes through the product's reader (symmetric banner: entries mirrored), the CSR cache, Tile_create,
    the plan and the whole-y check; the line names the file.  Written here: a symmetric pattern file (lower triangle only) and a general real file."""
    import json, subprocess, sys
    import scipy.sparse as sp
    from tilespmv_amd import generators as G
    m, n, rp, ci = G.tri_mesh(120, 120, shuffle=64)
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(m, n))
    L = sp.tril(A).tocoo()
    sym = tmp_path / "mesh_sym.mtx"
    with open(sym, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n")
        f.write("%d %d %d\n" % (m, n, L.nnz))
        for i, j in zip(L.row, L.col):
            f.write("%d %d\n" % (i + 1, j + 1))
    gen = tmp_path / "circuit_gen.mtx"
    m2, n2, rp2, ci2 = G.circuit_like(6000, seed=5)
    G.write_mtx(str(gen), m2, n2, rp2, ci2)
    for path, rows, nnz in ((sym, m, A.nnz), (gen, m2, len(ci2))):
        for rep in range(2):     # second run: CSR cache hit
            r = subprocess.run([sys.executable, "bench.py", "--workload", str(path), "--steps", "10", "--warmup", "3", "--no-extras", "--no-cpu-baseline", "--cache", str(tmp_path / "cache"),
                                "--full-json", str(tmp_path / "full.json")], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            assert line["config"]["source"].startswith("file:" + path.name) and line["check"].startswith("pass") and line["value"] > 0
            assert ("cache hit" in line["config"]["source"]) == (rep == 1)
            assert line["config"]["nnz"] == int(sp.csr_matrix((np.ones(nnz), (ci if path == sym else ci2), (rp if path == sym else rp2)), shape=(rows, rows))[: (rows // 16) * 16].nnz)


def test_auto_rules_pick_what_was_measured(torch_cuda):
    """The AUTO choices that round 2's sweep over unseen matrices corrected stay put: with the unit-stream
    kernel COO entries always run in-tile (also on uniform random matrices, where the old byte model chose the fallback); the
    matrix-core pass is used for dense tiles only when a tile-row holds several of them; entry-heavy shards use the merged
    lists, regular ones the per-strip walk; and every choice still gives the oracle's y."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api, generators as G
    cases = [("uniform random", G.random_uniform(200000, 200000, 8.0 / 200000, 1), {"coo_mode": api.COO_IN_TILE, "entry_mode": (1, 2)}),
             ("band hbw 12", G.band(200000, 12), {"coo_mode": api.COO_IN_TILE, "dense_mode": api.DENSE_VALU}),
             ("band hbw 40", G.band(400000, 40), {"coo_mode": api.COO_IN_TILE, "dense_mode": api.DENSE_MFMA}),
             ("5-pt Laplacian", G.laplacian5pt(512), {"coo_mode": api.COO_IN_TILE, "entry_mode": (0,)}),
             ("power-law", G.powerlaw(300000, seed=9), {"coo_mode": api.COO_IN_TILE, "entry_mode": (1, 2)})]
    for name, (m, n, rp, ci), want in cases:
        rowA, nnz = (m // 16) * 16, len(ci)
        vals, x = G.compat_values(nnz), G.compat_x(n)
        tp = api.Tile_create(rowA, n, nnz, rp, ci, vals)
        y, info = _gpu_y(torch_cuda, tp, rowA, n, nnz, x)
        assert np.array_equal(y, CpuImpl("oracle").csr_spmv(rowA, rp, ci, vals, x)), name
        for k, v in want.items():
            assert (info[k] in v) if isinstance(v, tuple) else (info[k] == v), (name, k, info[k], v)
        api.Tile_destroy(tp)
