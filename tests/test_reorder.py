"""Permuted-numbering helpers (round 6; include/tilespmv.h "Permuted-numbering plans"): reverse Cuthill-McKee on the host, B = P A P^T, and the CPU path on the
permuted matrix — P A P^T (P x) = P (A x), bit for bit on the reference driver's integer data (the oracle multiplies the permuted CSR; nothing here needs a GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

from tilespmv_amd import api, generators as G


def _bandwidth(A):
    c = A.tocoo()
    return int(np.abs(c.row - c.col).max()) if c.nnz else 0


@pytest.mark.parametrize("name,gen", [("tri shuffled", lambda: G.tri_mesh(60, 60, shuffle=256)), ("tet shuffled", lambda: G.tet_mesh(14, shuffle=128)),
                                      ("lap5", lambda: G.laplacian5pt(48)), ("power-law", lambda: G.powerlaw(3000, seed=2)),
                                      ("two components + isolated rows", lambda: _blocks())])
def test_rcm_is_a_permutation_and_narrows_the_band(name, gen):
    m, n, rp, ci = gen()
    assert m == n
    perm = api.reorder_rcm(n, rp, ci)
    assert sorted(perm.tolist()) == list(range(n))                       # a permutation
    assert np.array_equal(perm, api.reorder_rcm(n, rp, ci))              # deterministic
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    brp, bci, _ = api.csr_permute(n, rp, ci, np.ones(len(ci)), perm)
    B = sp.csr_matrix((np.ones(len(bci)), bci, brp), shape=(n, n))
    want = A[perm][:, perm]
    assert (B != want).nnz == 0                                          # B = P A P^T
    assert api.csr_bandwidth(n, brp, bci) == _bandwidth(B)
    ours, before = _bandwidth(B), _bandwidth(A)
    ps = reverse_cuthill_mckee((A + A.T).tocsr(), symmetric_mode=True)
    theirs = _bandwidth(A[ps][:, ps])
    if "shuffled" in name:
        assert ours < 0.7 * before, (name, before, ours)                 # shuffled windows: the band comes back (to about the mesh's natural one)
    assert ours <= max(1.25 * theirs, theirs + 8), (name, before, ours, theirs)   # not worse than scipy's ordering (ours starts from a pseudo-peripheral node)


def _blocks():
    a = G.laplacian5pt(12); b = G.tri_mesh(9, 9, shuffle=16)
    A = sp.block_diag([sp.csr_matrix((np.ones(len(a[3])), a[3], a[2]), shape=(a[0], a[1])), sp.csr_matrix((5, 5)),
                       sp.csr_matrix((np.ones(len(b[3])), b[3], b[2]), shape=(b[0], b[1]))]).tocsr()
    return A.shape[0], A.shape[1], A.indptr.astype(np.int32), A.indices.astype(np.int32)


def test_permute_keeps_halo_columns_sorts_rows_and_refuses_non_permutations():
    # a rank's [own | halo] index space: 4 own rows / columns, halo columns 4 .. 6
    rp = np.array([0, 3, 5, 7, 9], dtype=np.int32)
    ci = np.array([2, 0, 5, 1, 6, 3, 2, 4, 0], dtype=np.int32)
    v = np.arange(1, 10, dtype=np.float64)
    perm = np.array([2, 0, 3, 1], dtype=np.int32)                          # new row 0 = old row 2 ...
    brp, bci, bv = api.csr_permute(4, rp, ci, v, perm)
    inv = np.argsort(perm)
    assert brp.tolist() == [0, 2, 5, 7, 9]
    # rows 2, 0, 3, 1 of A; columns < 4 mapped through inv, halo columns (>= 4) kept; every row ascending by new column, values with their entries
    want = [sorted([(int(inv[3]), 6.0), (int(inv[2]), 7.0)]), sorted([(int(inv[2]), 1.0), (int(inv[0]), 2.0), (5, 3.0)]), sorted([(4, 8.0), (int(inv[0]), 9.0)]), sorted([(int(inv[1]), 4.0), (6, 5.0)])]
    assert bci.tolist() == [c for row in want for c, _ in row] and bv.tolist() == [v for row in want for _, v in row]
    with pytest.raises(ValueError):
        api.csr_permute(4, rp, ci, v, np.array([0, 0, 1, 2], dtype=np.int32))


def test_permuted_rows_ascend_so_that_dense_row_and_dense_col_tiles_stay_valid():
    """A 6-dof hex mesh is full of dense-row / dense-col tiles, whose packing takes the columns of a row to ascend (src/csr2tile.h:586,600-605 vs src/tilespmv_cpu.h:246,262).
    A first version of tilespmv_csr_permute kept A's entry order mapped through the permutation: the reference's own CPU tile path then returned 4,221 wrong rows on
    fem6s16_40.  Rows of B ascend; the tile path on B is exact."""
    from oracle.oracle import CpuImpl
    from tilespmv_amd.tile_matrix import to_dict
    m, n, rp, ci = G.fem_hex(9, 9, 9, 6, shuffle=16)
    rows = (m // 16) * 16
    A = sp.csr_matrix((np.ones(int(rp[rows])), ci[:int(rp[rows])], rp[:rows + 1]), shape=(rows, n))[:, :rows].tocsr()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    vals, x = G.compat_values(len(ci)), G.compat_x(rows)
    perm = api.reorder_rcm(rows, rp, ci)
    brp, bci, bv = api.csr_permute(rows, rp, ci, vals, perm)
    assert all(np.all(np.diff(bci[brp[i]:brp[i + 1]]) > 0) for i in range(rows))
    O = CpuImpl("oracle")
    y = O.csr_spmv(rows, rp, ci, vals, x)
    tm = api.Tile_create(rows, rows, len(bci), brp, bci, bv)
    fmt = np.bincount(to_dict(tm, rows)["Format"], minlength=7)
    assert fmt[5] > 0 and fmt[6] > 0            # dense-row and dense-col tiles are there
    r = api.tilespmv_cpu(tm, rows, rows, len(bci), brp, bci, bv, np.ascontiguousarray(x[perm]), O.csr_spmv(rows, brp, bci, bv, np.ascontiguousarray(x[perm])))
    api.Tile_destroy(tm)
    assert r["errcount"] == 0 and np.array_equal(r["y"][:rows], y[perm])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cpu_path_on_the_permuted_matrix_equals_the_permuted_result(dtype):
    """tilespmv_cpu on P A P^T with P x = P (tilespmv_cpu on A with x): exact on the reference driver's integer data (a permutation changes the order of the additions,
    integers do not care); checked through the oracle's CSR product as well."""
    from oracle.oracle import CpuImpl
    O = CpuImpl("oracle", dtype)
    m, n, rp, ci = G.tri_mesh(70, 70, shuffle=512)
    rows = (m // 16) * 16
    rp, ci = rp[:rows + 1].copy(), ci[:int(rp[rows])].copy()
    keep = ci < rows                                                      # square leading block
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(rows, n))[:, :rows].tocsr()
    rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    vals, x = G.compat_values(len(ci), dtype), G.compat_x(rows, dtype)
    perm = api.reorder_rcm(rows, rp, ci)
    brp, bci, bv = api.csr_permute(rows, rp, ci, vals, perm, dtype=dtype)
    y = O.csr_spmv(rows, rp, ci, vals, x)
    tm = api.Tile_create(rows, rows, len(bci), brp, bci, bv, dtype=dtype)
    yp = api.tilespmv_cpu(tm, rows, rows, len(bci), brp, bci, bv, np.ascontiguousarray(x[perm]), O.csr_spmv(rows, brp, bci, bv, np.ascontiguousarray(x[perm])))
    api.Tile_destroy(tm)
    assert yp["errcount"] == 0 and np.array_equal(yp["y"][:rows], y[perm])
