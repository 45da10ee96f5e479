"""Sharded-x SpMV with halo exchange + CG on CPU: 2 and 3 processes over gloo.  The local
multiply is the oracle's tile SpMV, so what is tested is the column renumbering, who-sends-what, the
all_to_all placement, the interior/edge row split and the solver loop — matvec bit-exact against the
1-process result on integer data, CG against scipy."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_dist_gloo import _OracleLocal  # noqa: E402


def _matrix(name):
    from tilespmv_amd import generators as G
    if name == "lap48":
        return G.laplacian5pt(48)            # 2304 rows
    if name == "band2048_40":
        return G.band(2048, 40)
    if name == "tri48s":                     # 2-D triangulation, nodes numbered in shuffled windows of 256 (2304 rows): what reordering is for
        return G.tri_mesh(48, 48, shuffle=256)
    if name == "powerlaw8k":
        m, n, rp, ci = G.powerlaw(8000)
        assert m == n
        return m, n, rp, ci
    raise KeyError(name)


def _spd_values(n, rp, ci):
    """Diagonally dominant symmetric values on a symmetric pattern: -1 off-diagonal, degree+1 on the diagonal."""
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = np.where(ci == rows, 0.0, -1.0)
    deg = np.bincount(rows, weights=(ci != rows).astype(np.float64), minlength=n)
    v[ci == rows] = deg[rows[ci == rows]] + 1.0
    return v


def _worker(rank, world, port, name, what, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tilespmv_amd import generators as G
    from tilespmv_amd.halo import HaloSpMV, cg
    m, n, rp, ci = _matrix(name)
    dtype = np.float64
    mk = lambda r, c, a, b, v: _OracleLocal(r, c, a, b, v, np.dtype(dtype))  # noqa: E731
    reorder = what.endswith("_rcm")
    if what.startswith("matvec"):
        vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
        A = HaloSpMV(rank, world, n, rp, ci, vals, dtype, make_local=mk, device="cpu", reorder=reorder)
        y = A.new_vector(-3.0)
        xin = A.to_plan_order(torch.from_numpy(np.concatenate([x[A.r0:A.r1], np.zeros(16)])))   # (a copy when the operator is not reordered)
        for _ in range(2):   # second call: stale halo values must be overwritten, not accumulated
            A.matvec(xin, y)
        res = A.from_plan_order(y)[:A.nloc].numpy().copy()
        meta = (A.nhalo, sum(A.send_splits), len(A.blocks)) + ((A.bandwidth[0], A.bandwidth[1]) if reorder else (0, 0))
    else:
        vals = _spd_values(n, rp, ci)
        rng = np.random.default_rng(7)
        bfull = rng.uniform(-1, 1, n)
        A = HaloSpMV(rank, world, n, rp, ci, vals, dtype, make_local=mk, device="cpu", reorder=reorder)
        b = A.new_vector(); b[:A.nloc] = torch.from_numpy(bfull[A.r0:A.r1].copy())
        x, it, rel = cg(A, b, tol=1e-11, maxiter=400)
        res = x[:A.nloc].numpy().copy()
        meta = (it, rel, len(A.blocks))
    parts = [None] * world
    dist.all_gather_object(parts, (A.r0, A.r1, res, meta))
    if rank == 0:
        full = np.zeros(n)
        for r0, r1, r, _ in parts:
            full[r0:r1] = r
        np.save(out, full)
        np.save(out + ".meta.npy", np.array([list(p[3]) for p in parts], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def _port(tag):
    return 31500 + (os.getpid() * 13 + hash(tag) % 1000) % 2000


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["lap48", "band2048_40", "powerlaw8k"])
def test_halo_matvec_matches_single_process(tmp_path, name, world):
    from oracle.oracle import CpuImpl
    from tilespmv_amd import generators as G
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(world, _port((name, world)), name, "matvec", out), nprocs=world, join=True)
    m, n, rp, ci = _matrix(name)
    vals, x = G.compat_values(len(ci)), G.compat_x(n)
    y1 = CpuImpl("oracle").csr_spmv(n, rp, ci, vals, x)
    assert np.array_equal(np.load(out), y1)
    meta = np.load(out + ".meta.npy")
    assert meta[:, 0].sum() == meta[:, 1].sum() and meta[:, 0].sum() > 0   # everything requested is sent
    if name == "lap48":
        # stencil: halo = one grid line per neighbour; the interior block is split off for overlap
        assert set(meta[:, 0]) <= {48.0, 96.0} and (meta[:, 2] >= 2).all()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_cg_converges_to_scipy_solution(tmp_path, world):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker, args=(world, _port(("cg", world)), "lap48", "cg", out), nprocs=world, join=True)
    m, n, rp, ci = _matrix("lap48")
    A = sp.csr_matrix((_spd_values(n, rp, ci), ci, rp), shape=(n, n))
    b = np.random.default_rng(7).uniform(-1, 1, n)
    xs = spla.spsolve(A.tocsc(), b)
    x = np.load(out)
    assert np.linalg.norm(x - xs) <= 1e-8 * np.linalg.norm(xs)
    meta = np.load(out + ".meta.npy")
    assert (meta[:, 0] == meta[0, 0]).all() and meta[0, 0] < 400 and (meta[:, 1] <= 1e-11).all()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_reordered_halo_operator_is_the_same_product_in_another_numbering(tmp_path, world):
    """HaloSpMV(reorder=True): every rank renumbers its own block by reverse Cuthill-McKee (rows, own columns, and what its neighbours fetch from it), vectors live in that
    numbering; permuted in at entry and out at exit the product is the 1-process result bit for bit on integer data — and the rank's band really got narrower."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd import generators as G
    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(world, _port(("rcm", world)), "tri48s", "matvec_rcm", out), nprocs=world, join=True)
    m, n, rp, ci = _matrix("tri48s")
    vals, x = G.compat_values(len(ci)), G.compat_x(n)
    assert np.array_equal(np.load(out), CpuImpl("oracle").csr_spmv(n, rp, ci, vals, x))
    meta = np.load(out + ".meta.npy")
    assert (meta[:, 4] < 0.7 * meta[:, 3]).all(), meta          # bandwidth of the own block: after < before
    assert meta[:, 0].sum() == meta[:, 1].sum()


def test_cg_on_a_reordered_operator_permutes_once_at_entry_and_exit(tmp_path):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker, args=(2, _port(("cg_rcm", 2)), "tri48s", "cg_rcm", out), nprocs=2, join=True)
    m, n, rp, ci = _matrix("tri48s")
    A = sp.csr_matrix((_spd_values(n, rp, ci), ci, rp), shape=(n, n))
    b = np.random.default_rng(7).uniform(-1, 1, n)
    xs = spla.spsolve(A.tocsc(), b)
    x = np.load(out)
    assert np.linalg.norm(x - xs) <= 1e-8 * np.linalg.norm(xs)


def test_halo_single_rank_is_plain_spmv():
    from oracle.oracle import CpuImpl
    from tilespmv_amd import generators as G
    from tilespmv_amd.halo import HaloSpMV
    m, n, rp, ci = _matrix("lap48")
    vals, x = G.compat_values(len(ci)), G.compat_x(n)
    A = HaloSpMV(0, 1, n, rp, ci, vals, make_local=lambda r, c, a, b, v: _OracleLocal(r, c, a, b, v, np.dtype(np.float64)), device="cpu")
    assert A.nhalo == 0 and len(A.blocks) == 1 and A.halo_bytes() == 0
    y = A.matvec(torch.from_numpy(x.copy()), A.new_vector())
    assert np.array_equal(y[:n].numpy(), CpuImpl("oracle").csr_spmv(n, rp, ci, vals, x))
    with pytest.raises(ValueError):
        HaloSpMV(0, 1, 1000, rp, ci, vals, device="cpu")
