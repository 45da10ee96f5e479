"""Permuted-numbering plans on the GPU (round 6): the HIP plan of P A P^T, fed P x through tilespmv_permute_vector, gives P (A x) — the oracle's result on the same
permuted CSR, bit for bit on the reference driver's integer data — and the un-permuted y is the plain product.  HaloSpMV(reorder=True) + cg on one GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from tilespmv_amd import api, generators as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _square(gen):
    m, n, rp, ci = gen
    rows = (m // 16) * 16
    A = sp.csr_matrix((np.ones(int(rp[rows])), ci[:int(rp[rows])], rp[:rows + 1]), shape=(rows, n))[:, :rows].tocsr()
    A.sum_duplicates()   # a simple graph: with repeated (i, j) entries a renumbering that packs hub rows and popular columns into one 16 x 16 tile can put more than 255 entries
    return rows, A.indptr.astype(np.int32), A.indices.astype(np.int32)   # there, which the reference's unsigned-char tile counters do not hold (SURVEY S8c "input hazards")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,gen", [("tri300s1024", lambda: G.tri_mesh(300, 300, shuffle=1024)), ("tet40s256", lambda: G.tet_mesh(40, shuffle=256)),
                                      ("fem3s64_16", lambda: G.fem_hex(16, 16, 16, 3, shuffle=64)), ("bandrand40k", lambda: G.band_plus_random(40000, 4, 3, 5)),
                                      ("fem6s16_12 (dense-row / dense-col tiles)", lambda: G.fem_hex(12, 12, 12, 6, shuffle=16))])
def test_plan_of_the_permuted_matrix_bit_exact(torch_cuda, name, gen, dtype):
    from oracle.oracle import CpuImpl
    torch = torch_cuda
    O = CpuImpl("oracle", dtype)
    n, rp, ci = _square(gen())
    vals, x = G.compat_values(len(ci), dtype), G.compat_x(n, dtype)
    perm = api.reorder_rcm(n, rp, ci)
    brp, bci, bv = api.csr_permute(n, rp, ci, vals, perm, dtype=dtype)
    y = O.csr_spmv(n, rp, ci, vals, x)
    want_p = O.spmv(O.tile_create(n, n, len(bci), brp, bci, bv), n, n, len(bci), brp, bci, bv, np.ascontiguousarray(x[perm]))["y"]   # the oracle on the SAME permuted CSR
    assert np.array_equal(want_p, y[perm])
    st = torch.cuda.current_stream().cuda_stream
    pd = torch.from_numpy(perm).cuda()
    xd = torch.from_numpy(x).cuda()
    xp = torch.zeros(n + 16, dtype=xd.dtype, device="cuda"); yp = torch.full((n + 16,), 777.0, dtype=xd.dtype, device="cuda"); yo = torch.full((n + 16,), 777.0, dtype=xd.dtype, device="cuda")
    for how in ("host tiles", "device build"):
        if how == "host tiles":
            tm = api.Tile_create(n, n, len(bci), brp, bci, bv, dtype=dtype)
            plan = api.Plan(tm, n, n, len(bci)); api.Tile_destroy(tm)
        else:
            plan = api.Plan.from_csr(n, n, len(bci), brp, bci, bv, dtype=dtype)
        api.permute_vector(xd.data_ptr(), xp.data_ptr(), pd.data_ptr(), n, scatter=False, stream=st, dtype=dtype)     # x into plan order
        plan.spmv(xp.data_ptr(), yp.data_ptr(), st)
        api.permute_vector(yp.data_ptr(), yo.data_ptr(), pd.data_ptr(), n, scatter=True, stream=st, dtype=dtype)      # y back
        torch.cuda.synchronize()
        assert np.array_equal(xp.cpu().numpy()[:n], x[perm])
        assert np.array_equal(yp.cpu().numpy()[:n], want_p), (name, how)
        assert np.array_equal(yo.cpu().numpy()[:n], y), (name, how)
        assert float(yo[n]) == 777.0 and float(yp[n + 15]) == 777.0        # nothing past the end
        plan.close()


def test_reordered_halo_operator_and_cg_on_one_gpu(torch_cuda):
    import scipy.sparse.linalg as spla
    from oracle.oracle import CpuImpl
    from tilespmv_amd.halo import HaloSpMV, cg
    torch = torch_cuda
    n, rp, ci = _square(G.tri_mesh(200, 200, shuffle=1024))
    # the product: integer data, bit for bit
    vals, x = G.compat_values(len(ci)), G.compat_x(n)
    A = HaloSpMV(0, 1, n, rp, ci, vals, reorder=True)
    assert A.bandwidth[1] < 0.5 * A.bandwidth[0]
    xin = torch.zeros(n + 16, dtype=torch.float64, device="cuda"); xin[:n] = torch.from_numpy(x).cuda()
    y = A.from_plan_order(A.matvec(A.to_plan_order(xin), A.new_vector()))
    torch.cuda.synchronize()
    assert np.array_equal(y[:n].cpu().numpy(), CpuImpl("oracle").csr_spmv(n, rp, ci, vals, x))
    A.close()
    # the solver: SPD values on the (symmetric) mesh pattern, solution against scipy
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = np.where(ci == rows, 0.0, -1.0)
    deg = np.bincount(rows, weights=(ci != rows).astype(np.float64), minlength=n)
    v[ci == rows] = deg[rows[ci == rows]] + 1.0
    b = np.random.default_rng(3).uniform(-1, 1, n)
    xs = spla.spsolve(sp.csr_matrix((v, ci, rp), shape=(n, n)).tocsc(), b)
    for reorder in (False, True):
        A = HaloSpMV(0, 1, n, rp, ci, v, reorder=reorder)
        bd = A.new_vector(); bd[:n] = torch.from_numpy(b).cuda()
        xsol, it, rel = cg(A, bd, tol=1e-11, maxiter=500)
        torch.cuda.synchronize()
        assert rel <= 1e-11 and np.linalg.norm(xsol[:n].cpu().numpy() - xs) <= 1e-8 * np.linalg.norm(xs), (reorder, it, rel)
        A.close()
