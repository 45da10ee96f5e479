"""Deterministic synthetic CSR matrices for the BASELINE configs and the parity tests.

No SuiteSparse file ships with the reference and there is no network, so each
BASELINE config has a generator with the same shape / nnz class; a real ``.mtx`` is used
instead when ``$TILESPMV_MATRIX_DIR/<name>.mtx`` exists (see ``bench.py``).

All generators return ``(rows, cols, rowptr[int32], colidx[int32])``; values are assigned by
the caller: ``compat_values`` reproduces the reference driver's synthetic data
(``val[i] = i % 10``, ``x[i] = i % 10``: reference src/main.cu:68-69, :93-97).
"""
import numpy as np


def compat_values(nnz, dtype=np.float64, first=0):
    """val[i] = i % 10 over the GLOBAL nonzero index i; ``first`` = index of this block's first nonzero."""
    return ((np.arange(nnz, dtype=np.int64) + int(first)) % 10).astype(dtype)


def compat_x(n, dtype=np.float64):
    return (np.arange(n, dtype=np.int64) % 10).astype(dtype)


REAL_SEED = 12345


def real_values(nnz, dtype=np.float64, first=0):
    """SURVEY.md S8(d) data mode (ii): vals ~ U(-1, 1), seed 12345 — nonzeros [first, first + nnz) of ONE global stream
    (PCG64 advanced to ``first``: a rank generates only its own block and still sees the values a one-rank run would)."""
    bg = np.random.PCG64(REAL_SEED)
    bg.advance(int(first))
    return np.random.Generator(bg).uniform(-1.0, 1.0, int(nnz)).astype(dtype)


def real_x(n, nnz_total, dtype=np.float64):
    """x ~ U(-1, 1) from the same stream, right behind the ``nnz_total`` values (the order tests/cases.values_for uses)."""
    bg = np.random.PCG64(REAL_SEED)
    bg.advance(int(nnz_total))
    return np.random.Generator(bg).uniform(-1.0, 1.0, int(n)).astype(dtype)


def _from_mask(cand, mask):
    """cand/mask: (rows, k) candidate columns in the wanted per-row order + validity."""
    counts = mask.sum(axis=1, dtype=np.int64)
    rowptr = np.zeros(cand.shape[0] + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    colidx = cand[mask].astype(np.int32)
    assert rowptr[-1] < 2**31
    return rowptr.astype(np.int32), colidx


def _block(cand_fn, N, rows):
    """Rows [r0, r1) of a stencil matrix: (N, N, rowptr rebased to 0, colidx) — only that block is ever materialised."""
    r0, r1 = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    idx = np.arange(r0, r1, dtype=np.int64)
    cand, mask = cand_fn(idx)
    rowptr, colidx = _from_mask(cand, mask)
    return N, N, rowptr, colidx


def laplacian5pt(n, rows=None):
    """5-point stencil on an n x n row-major grid; per-row order up,left,centre,right,down
    (SURVEY.md §8c known-answer generator; BASELINE config 4 is n=4096).  ``rows=(r0, r1)``: only that row block
    (global column ids, row pointer rebased) — what one rank of a sharded run generates."""
    def cand_fn(idx):
        i, j = idx // n, idx % n
        return (np.stack([idx - n, idx - 1, idx, idx + 1, idx + n], axis=1),
                np.stack([i > 0, j > 0, np.ones(idx.size, bool), j < n - 1, i < n - 1], axis=1))
    return _block(cand_fn, n * n, rows)


def laplacian5pt_rowptr(n):
    """Row pointer of laplacian5pt(n) alone (int64): cheap, lets every rank cut the same nnz-balanced partition."""
    idx = np.arange(n * n, dtype=np.int64)
    i, j = idx // n, idx % n
    deg = 5 - (i == 0) - (i == n - 1) - (j == 0) - (j == n - 1)
    rp = np.zeros(n * n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    return rp


def laplacian7pt(n, rows=None):
    """7-point stencil on an n^3 grid (row-major), per-row order by ascending column.  ``rows``: see laplacian5pt."""
    def cand_fn(idx):
        k, j, i = idx // (n * n), (idx // n) % n, idx % n
        return (np.stack([idx - n * n, idx - n, idx - 1, idx, idx + 1, idx + n, idx + n * n], axis=1),
                np.stack([k > 0, j > 0, i > 0, np.ones(idx.size, bool), i < n - 1, j < n - 1, k < n - 1], axis=1))
    return _block(cand_fn, n * n * n, rows)


def laplacian7pt_rowptr(n):
    idx = np.arange(n * n * n, dtype=np.int64)
    k, j, i = idx // (n * n), (idx // n) % n, idx % n
    deg = 7 - (k == 0) - (k == n - 1) - (j == 0) - (j == n - 1) - (i == 0) - (i == n - 1)
    rp = np.zeros(n * n * n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    return rp


def band(n, hbw, ncols=None):
    """Full band: row r holds columns r-hbw..r+hbw (clipped), ascending."""
    ncols = n if ncols is None else ncols
    r = np.arange(n, dtype=np.int64)[:, None]
    cand = r + np.arange(-hbw, hbw + 1, dtype=np.int64)[None, :]
    mask = (cand >= 0) & (cand < ncols)
    rowptr, colidx = _from_mask(cand, mask)
    return n, ncols, rowptr, colidx


def from_coo(rows, cols, ri, ci, sort_cols=True):
    """CSR from coordinate lists (duplicates removed); columns ascending inside a row."""
    ri = np.asarray(ri, dtype=np.int64); ci = np.asarray(ci, dtype=np.int64)
    key = np.unique(ri * cols + ci)
    ri, ci = key // cols, key % cols
    rowptr = np.zeros(rows + 1, dtype=np.int64)
    np.add.at(rowptr, ri + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rows, cols, rowptr.astype(np.int32), ci.astype(np.int32)


def retarget_nnz(rows, cols, rowptr, colidx, target_nnz, seed=0, near=40):
    """Same structure with exactly ``target_nnz`` nonzeros: a surplus is removed uniformly at random, a deficit is
    filled with new off-diagonals within +-``near`` of the diagonal (deterministic for a given seed).  Used to size the
    SuiteSparse stand-ins like the matrices they stand for (reference src/external/CSR5_cuda/2757-matrix.csv)."""
    rng = np.random.default_rng(seed)
    ri = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rowptr))
    key = ri * cols + np.asarray(colidx, dtype=np.int64)          # sorted and unique: rows ascending, columns ascending inside
    if len(key) > target_nnz:
        keep = np.sort(rng.choice(len(key), target_nnz, replace=False))
        key = key[keep]
    while len(key) < target_nnz:
        need = target_nnz - len(key)
        r = rng.integers(0, rows, int(need * 1.3) + 16)
        c = np.clip(r + rng.integers(-near, near + 1, r.size), 0, cols - 1)
        new = np.setdiff1d(np.unique(r * cols + c), key, assume_unique=True)
        if new.size > need:
            new = np.sort(rng.choice(new, need, replace=False))
        key = np.union1d(key, new)
    ri, ci = key // cols, key % cols
    rp = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(ri, minlength=rows), out=rp[1:])
    return rows, cols, rp.astype(np.int32), ci.astype(np.int32)


def random_uniform(rows, cols, density, seed):
    rng = np.random.default_rng(seed)
    nnz = int(rows * cols * density)
    return from_coo(rows, cols, rng.integers(0, rows, nnz), rng.integers(0, cols, nnz))


def uniform_per_row(rows, cols, per_row, seed):
    """``per_row`` uniformly random columns in every row (duplicates removed): the fully irregular end of the HBM-resident
    test population (the authors' 2,757-matrix sweep is dominated by irregular matrices)."""
    rng = np.random.default_rng(seed)
    ri = np.repeat(np.arange(rows, dtype=np.int64), per_row)
    return from_coo(rows, cols, ri, rng.integers(0, cols, ri.size))


def band_plus_random(n, hbw, extra, seed):
    """Full band of half-bandwidth ``hbw`` plus ``extra`` uniformly random entries per row on average: a regular part
    (units) with scattered fill (one-entry COO tiles) — what many "structured + coupling" SuiteSparse matrices look like."""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n, dtype=np.int64), 2 * hbw + 1)
    c = r + np.tile(np.arange(-hbw, hbw + 1, dtype=np.int64), n)
    ok = (c >= 0) & (c < n)
    rr = rng.integers(0, n, extra * n); cc = rng.integers(0, n, extra * n)
    return from_coo(n, n, np.concatenate([r[ok], rr]), np.concatenate([c[ok], cc]))


def rmat(scale, edge_factor, seed, a=0.57, b=0.19, c=0.19):
    """R-MAT (Graph500 parameters) with 2^scale vertices and edge_factor * 2^scale edges before duplicate removal."""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = edge_factor * n
    ri = np.zeros(m, np.int64); ci = np.zeros(m, np.int64)
    for _ in range(scale):
        u = rng.random(m)
        rbit = (u >= a + b).astype(np.int64)
        cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.int64)
        ri = (ri << 1) | rbit; ci = (ci << 1) | cbit
    return from_coo(n, n, ri, ci)


def block_diag_plus_sparse(nb, bs, extra, seed, fill=0.6):
    """``nb`` diagonal blocks of ``bs`` x ``bs`` at ``fill`` density + ``extra`` random entries per row."""
    rng = np.random.default_rng(seed)
    n = nb * bs
    keep = rng.random((nb, bs, bs)) < fill
    b, i, j = np.nonzero(keep)
    rr = rng.integers(0, n, extra * n); cc = rng.integers(0, n, extra * n)
    return from_coo(n, n, np.concatenate([b * bs + i, rr]), np.concatenate([b * bs + j, cc]))


def all_formats(nblk=12, seed=7, cols_pad=0):
    """Small matrix whose 16x16 tiles are built on purpose to hit every selection rule of
    reference src/csr2tile.h:143-325: dense, COO, dense-row, dense-col, ELL, CSR and (only when
    the HYB rule is enabled) HYB.  ``cols_pad`` > 0 makes the last tile column partial."""
    rng = np.random.default_rng(seed)
    rows = 16 * nblk
    cols = 16 * nblk - (16 - cols_pad if cols_pad else 0)
    R, Cc = [], []

    def put(bi, bj, lr, lc):
        lr = np.asarray(lr); lc = np.asarray(lc)
        ok = (16 * bj + lc) < cols
        R.extend((16 * bi + lr[ok]).tolist()); Cc.extend((16 * bj + lc[ok]).tolist())

    g = np.arange(16)
    for bi in range(nblk):
        kinds = ["dense", "coo", "dnsrow", "dnscol", "ell", "csr", "hyb", "ell1", "coo1", "dense75"]
        if bi % 3 == 1:
            kinds = kinds[bi % 4::3]  # short tile-rows (<= 4 tiles: not split by the schedule)
        for k, kind in enumerate(kinds):
            bj = (bi + k) % nblk if bi % 2 == 0 else (bi + nblk - k) % nblk
            if kind == "dense":
                lr, lc = np.meshgrid(g, g, indexing="ij"); put(bi, bj, lr.ravel(), lc.ravel())
            elif kind == "dense75":
                m = rng.random((16, 16)) < 0.85
                lr, lc = np.nonzero(m); put(bi, bj, lr, lc)
            elif kind == "coo":
                n = int(rng.integers(1, 13)); p = rng.choice(256, n, replace=False); put(bi, bj, p // 16, p % 16)
            elif kind == "coo1":
                put(bi, bj, np.array([int(rng.integers(16))]), np.array([int(rng.integers(16))]))
            elif kind == "dnsrow":
                rs = rng.choice(16, int(rng.integers(1, 6)), replace=False)
                lr, lc = np.meshgrid(rs, g, indexing="ij"); put(bi, bj, lr.ravel(), lc.ravel())
            elif kind == "dnscol":
                cs = rng.choice(16, int(rng.integers(1, 6)), replace=False)
                lr, lc = np.meshgrid(g, cs, indexing="ij"); put(bi, bj, lr.ravel(), lc.ravel())
            elif kind == "ell":
                w = int(rng.integers(2, 7))
                for r in range(16):
                    put(bi, bj, np.full(w, r), rng.choice(16, w, replace=False))
            elif kind == "ell1":
                put(bi, bj, g, (g + bi) % 16)
            elif kind == "csr":
                for r in range(16):
                    w = int(rng.integers(0, 9))
                    if w: put(bi, bj, np.full(w, r), rng.choice(16, w, replace=False))
            elif kind == "hyb":  # ten 1-entry rows + one long row: variation >= 1, remainder <= 4
                rs = rng.choice(16, 11, replace=False)
                for r in rs[:10]:
                    put(bi, bj, np.array([r]), np.array([int(rng.integers(16))]))
                put(bi, bj, np.full(4, rs[10]), rng.choice(16, 4, replace=False))
    return from_coo(rows, cols, R, Cc)


def powerlaw(n, seed=2, alpha=2.1, maxdeg=4700, near=30):
    """webbase-1M stand-in (BASELINE config 3): Zipf(alpha) out-degrees, half the targets near
    the diagonal (+-near), half Pareto-popular columns; n x n."""
    rng = np.random.default_rng(seed)
    deg = np.minimum(rng.zipf(alpha, n), maxdeg).astype(np.int64)
    tot = int(deg.sum())
    ri = np.repeat(np.arange(n, dtype=np.int64), deg)
    local = rng.random(tot) < 0.5
    off = rng.integers(-near, near + 1, tot)
    pop = np.minimum((rng.pareto(1.2, tot) * (n / 2000.0)).astype(np.int64), n - 1)
    ci = np.where(local, np.clip(ri + off, 0, n - 1), pop)
    return from_coo(n, n, ri, ci)


def circuit_like(n, seed=1, avg_off=4.6, ndense=6):
    """scircuit stand-in (BASELINE config 2): diagonal + a few local off-diagonals per row + a
    handful of dense rows/columns, plus injected 16x16 block patterns so that every tile format
    occurs."""
    rng = np.random.default_rng(seed)
    k = rng.poisson(avg_off, n).astype(np.int64)
    ri = np.repeat(np.arange(n, dtype=np.int64), k)
    span = np.where(rng.random(ri.size) < 0.8, 40, n)
    ci = np.clip(ri + (rng.standard_normal(ri.size) * span).astype(np.int64), 0, n - 1)
    R = [np.arange(n, dtype=np.int64), ri]; Cc = [np.arange(n, dtype=np.int64), ci]
    hubs = rng.choice(n, ndense, replace=False)
    for h in hubs:
        m = rng.choice(n, n // 400, replace=False)
        R += [np.full(m.size, h), m]; Cc += [m, np.full(m.size, h)]
    g = np.arange(16)
    nb = n // 16
    for b in rng.choice(nb - 1, min(200, nb - 1), replace=False):
        kind = int(rng.integers(0, 5)); bj = int(rng.integers(0, nb - 1))
        if kind == 0: lr, lc = np.meshgrid(g, g, indexing="ij")
        elif kind == 1: lr, lc = np.meshgrid(rng.choice(16, 3, replace=False), g, indexing="ij")
        elif kind == 2: lr, lc = np.meshgrid(g, rng.choice(16, 3, replace=False), indexing="ij")
        elif kind == 3: lr, lc = np.repeat(g, 3), rng.integers(0, 16, 48)
        else:
            lr = np.concatenate([g[:10], np.full(4, 12)]); lc = rng.integers(0, 16, 14)
        R.append(16 * b + lr.ravel()); Cc.append(16 * bj + lc.ravel())
    return from_coo(n, n, np.concatenate(R), np.concatenate(Cc))


def kkt_like(g, seed=5):
    """Small KKT-like test matrix (tests/cases.py ``kkt12``): symmetric [[H, A^T],[A, 0]] on a g^3 grid — H is a
    27-point stencil on g^3 unknowns, A couples each of the g^3 constraints to a 7-point neighbourhood."""
    n = g * g * g
    idx = np.arange(n, dtype=np.int64)
    z, y, x = idx // (g * g), (idx // g) % g, idx % g
    Rs, Cs = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = (z + dz >= 0) & (z + dz < g) & (y + dy >= 0) & (y + dy < g) & (x + dx >= 0) & (x + dx < g)
                Rs.append(idx[ok]); Cs.append(idx[ok] + (dz * g + dy) * g + dx)
    for dz, dy, dx in ((0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        ok = (z + dz >= 0) & (z + dz < g) & (y + dy >= 0) & (y + dy < g) & (x + dx >= 0) & (x + dx < g)
        a_r = n + idx[ok]; a_c = idx[ok] + (dz * g + dy) * g + dx
        Rs += [a_r, a_c]; Cs += [a_c, a_r]
    R = np.concatenate(Rs); Cc = np.concatenate(Cs)
    order = np.lexsort((Cc, R))
    R, Cc = R[order], Cc[order]
    rowptr = np.zeros(2 * n + 1, dtype=np.int64)
    np.cumsum(np.bincount(R, minlength=2 * n), out=rowptr[1:])
    return 2 * n, 2 * n, rowptr.astype(np.int32), Cc.astype(np.int32)


def _fem_node_order(nn, shuffle, seed):
    """new index of every node: identity, or a random permutation inside consecutive windows of ``shuffle`` nodes."""
    if not shuffle or shuffle <= 1:
        return None
    rng = np.random.default_rng(seed)
    win = np.arange(nn, dtype=np.int64) // shuffle
    order = np.lexsort((rng.random(nn), win))      # inside a window: by random key
    new_of_old = np.empty(nn, dtype=np.int64)
    new_of_old[order] = np.arange(nn, dtype=np.int64)   # (windows are contiguous ranges, so positions stay inside the window)
    return new_of_old


def mesh_matrix(nx, ny, nz, dof, offs, shuffle=0, seed=11, rows=None, rowptr_only=False):
    """Sparsity of a mesh discretisation: nodes on an nx x ny x nz grid, ``dof`` unknowns per node, every node coupled to the nodes at the offsets ``offs`` (dz, dy, dx; the
    node itself included if (0, 0, 0) is listed) with a full dof x dof block, columns ascending.  Natural (x fastest) node order, or, with ``shuffle`` = w, nodes renumbered
    at random inside windows of w consecutive nodes (a mediocre ordering: locality kept at the scale of the window, none inside it).  ``rows=(r0, r1)``: only that row
    block (global columns, row pointer rebased); ``rowptr_only``: the int64 row pointer alone."""
    nn = nx * ny * nz
    N = nn * dof
    new_of_old = _fem_node_order(nn, shuffle, seed)
    if new_of_old is None:
        old_of_new = None
    else:
        old_of_new = np.empty(nn, dtype=np.int64)
        old_of_new[new_of_old] = np.arange(nn, dtype=np.int64)
    offs = sorted(offs)
    K = len(offs)

    def node_nb(nodes_new):
        """(len, K) neighbour nodes (new numbering, ascending, invalid = nn at the end) of the given nodes (new numbering)."""
        old = nodes_new if old_of_new is None else old_of_new[nodes_new]
        z, y, x = old // (nx * ny), (old // nx) % ny, old % nx
        nb = np.empty((len(old), K), dtype=np.int64)
        for k, (dz, dy, dx) in enumerate(offs):
            ok = (z + dz >= 0) & (z + dz < nz) & (y + dy >= 0) & (y + dy < ny) & (x + dx >= 0) & (x + dx < nx)
            q = old + (dz * ny + dy) * nx + dx
            if new_of_old is not None:
                q = new_of_old[np.where(ok, q, 0)]
            nb[:, k] = np.where(ok, q, nn)
        if new_of_old is not None:
            nb.sort(axis=1)
        return nb

    if rowptr_only:
        idx = np.arange(nn, dtype=np.int64)
        old = idx if old_of_new is None else old_of_new
        z, y, x = old // (nx * ny), (old // nx) % ny, old % nx
        cnt = np.zeros(nn, dtype=np.int64)
        for dz, dy, dx in offs:
            cnt += (z + dz >= 0) & (z + dz < nz) & (y + dy >= 0) & (y + dy < ny) & (x + dx >= 0) & (x + dx < nx)
        rp = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(np.repeat(cnt * dof, dof), out=rp[1:])
        return rp
    r0, r1 = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    n0, n1 = r0 // dof, (r1 + dof - 1) // dof
    nb = node_nb(np.arange(n0, n1, dtype=np.int64))
    valid = nb < nn
    cols_node = (nb[:, :, None] * dof + np.arange(dof, dtype=np.int64)[None, None, :]).reshape(len(nb), K * dof)
    mask_node = np.repeat(valid, dof, axis=1)
    del nb, valid
    # the dof rows of a node share its column list
    rsel = np.arange(r0, r1, dtype=np.int64) // dof - n0
    counts = mask_node.sum(axis=1, dtype=np.int64)[rsel]
    rowptr = np.zeros(r1 - r0 + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    assert rowptr[-1] < 2**31
    if r0 % dof == 0 and r1 % dof == 0:
        # whole nodes: repeat every node's compacted column list dof times without materialising the (rows, K dof) candidate array
        flat = cols_node[mask_node].astype(np.int32)
        per_node = mask_node.sum(axis=1, dtype=np.int64)
        starts = np.zeros(len(per_node) + 1, dtype=np.int64)
        np.cumsum(per_node, out=starts[1:])
        del cols_node, mask_node
        colidx = np.empty(int(rowptr[-1]), dtype=np.int32)
        # rows d, d + dof, ... of the block take the node lists in order: row-major positions rowptr[d::dof]
        for d in range(dof):
            dst = rowptr[d:-1:dof]
            idx = np.repeat(dst - starts[:-1], per_node) + np.arange(len(flat), dtype=np.int64)
            colidx[idx] = flat
        return N, N, rowptr.astype(np.int32), colidx
    colidx = cols_node[rsel][mask_node[rsel]].astype(np.int32)
    return N, N, rowptr.astype(np.int32), colidx

OFFS_27 = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
OFFS_9 = [(0, dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
OFFS_TRI7 = [(0, 0, 0), (0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (0, 1, 1), (0, -1, -1)]                      # structured triangulation of the plane: 6 neighbours + the node
OFFS_TET15 = sorted(set([(0, 0, 0)] + [s for a in ((0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)) for s in (a, (-a[0], -a[1], -a[2]))]))   # Kuhn subdivision of the cubes: 14 neighbours + the node


def fem_hex(nx, ny, nz, dof, shuffle=0, seed=11, rows=None, rowptr_only=False):
    """Structural-FEM-like matrix (the largest class among the >= 10 M-nnz matrices of the reference's sweep list,
    src/external/CSR5_cuda/2757-matrix.csv: audikw_1 :1252, ldoor :1268, bone010 :1453, Flan_1565 :2544 ...): trilinear hexahedra
    on an nx x ny x nz grid of nodes, ``dof`` unknowns per node, every node coupled to its 27-point neighbourhood with a
    full dof x dof block — 81 (dof 3) / 162 (dof 6) nonzeros in an interior row.  nz = 1 gives the 9-point quadrilateral
    mesh of a shell model (af_shell10 :1586: 34.9 nonzeros per row = 9 neighbours x 4 unknowns).  See ``mesh_matrix``."""
    return mesh_matrix(nx, ny, nz, dof, OFFS_27 if nz > 1 else OFFS_9, shuffle, seed, rows, rowptr_only)


def tri_mesh(nx, ny, shuffle=0, seed=12, rows=None, rowptr_only=False):
    """2-D triangulation (delaunay_n2x-like, :2476-2479: 6.0 nonzeros per row): every node of an nx x ny grid coupled to its six neighbours in a structured triangulation, plus itself."""
    return mesh_matrix(nx, ny, 1, 1, OFFS_TRI7, shuffle, seed, rows, rowptr_only)


def tet_mesh(n, dof=1, shuffle=0, seed=13, rows=None, rowptr_only=False):
    """3-D tetrahedral mesh (CFD / electromagnetics class, 12-20 nonzeros per row): Kuhn subdivision of an n^3 grid of cubes, every node coupled to its 14 neighbours and itself."""
    return mesh_matrix(n, n, n, dof, OFFS_TET15, shuffle, seed, rows, rowptr_only)


def road_like(nx, ny, keep=0.62, shuffle=0, seed=14):
    """Road-network-like graph (*_osm, road_usa :2509-2514: 2.1-2.4 nonzeros per row, no diagonal): the 4-neighbour grid graph of nx x ny junctions with each (undirected) edge kept
    with probability ``keep``; natural order or window-shuffled like the meshes."""
    rng = np.random.default_rng(seed)
    nn = nx * ny
    idx = np.arange(nn, dtype=np.int64)
    y, x = idx // nx, idx % nx
    right = (x < nx - 1) & (rng.random(nn) < keep); down = (y < ny - 1) & (rng.random(nn) < keep)
    a = np.concatenate([idx[right], idx[down]]); b = np.concatenate([idx[right] + 1, idx[down] + nx])
    new_of_old = _fem_node_order(nn, shuffle, seed)
    if new_of_old is not None:
        a, b = new_of_old[a], new_of_old[b]
    return from_coo(nn, nn, np.concatenate([a, b]), np.concatenate([b, a]))


NLPKKT160_ROWS, NLPKKT160_NNZ = 8345600, 229518112   # reference src/external/CSR5_cuda/2757-matrix.csv:1903


def nlpkkt_like(g=160, target_nnz=None, seed=5):
    """nlpkkt160 stand-in (BASELINE config 5), sized like the real matrix: symmetric [[H, A^T],[A, 0]] with
    g*g*(g+6) primal unknowns on a (g+6) x g x g grid (H = 27-point stencil) and g^3 constraints, each coupled to a
    15-point neighbourhood (7-point star + 8 corners) of the primal cell three planes up.  g = 160 gives exactly
    8,345,600 rows (= 2*160^3 + 6*160^2, the row count of nlpkkt160) and, with ``target_nnz`` (default: the real
    229,518,112), exactly that many nonzeros: the surplus is taken off one corner coupling, symmetrically, for a hashed
    subset of the constraints.  Built row by row from ascending candidate columns: no sort of the 2.3e8 entries."""
    if target_nnz is None and g == 160:
        target_nnz = NLPKKT160_NNZ
    gz = g + 6
    n1, n2 = gz * g * g, g * g * g
    st27 = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    st15 = sorted([(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)] +
                  [(a, b, c) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)])
    thin = (1, 1, 1)                                        # the coupling that absorbs the surplus
    # constraint c = (z, y, x) couples to primal (z + 3 + dz, y + dy, x + dx): always inside in z, clipped in y and x
    c_idx = np.arange(n2, dtype=np.int64)
    cz, cy, cx = c_idx // (g * g), (c_idx // g) % g, c_idx % g
    hashv = (c_idx * 2654435761) % (1 << 32)

    def a_valid(dz, dy, dx):
        return (cy + dy >= 0) & (cy + dy < g) & (cx + dx >= 0) & (cx + dx < g)

    p_idx = np.arange(n1, dtype=np.int64)
    pz, py, px = p_idx // (g * g), (p_idx // g) % g, p_idx % g
    h_masks = [((pz + dz >= 0) & (pz + dz < gz) & (py + dy >= 0) & (py + dy < g) & (px + dx >= 0) & (px + dx < g)) for dz, dy, dx in st27]
    nnz_h = int(sum(int(m.sum()) for m in h_masks))
    nnz_a = int(sum(int(a_valid(*o).sum()) for o in st15))
    drop = np.zeros(n2, dtype=bool)
    if target_nnz is not None:
        surplus = nnz_h + 2 * nnz_a - target_nnz
        assert surplus >= 0 and surplus % 2 == 0, (nnz_h, nnz_a, target_nnz)
        cand = np.nonzero(a_valid(*thin))[0]
        assert surplus // 2 <= cand.size
        order = cand[np.argsort(hashv[cand], kind="stable")]
        drop[order[:surplus // 2]] = True
    # ---- primal rows: H columns ascending, then the A^T columns (n1 + constraint) ascending
    cand_cols, cand_mask = [], []
    for (dz, dy, dx), m in zip(st27, h_masks):
        cand_cols.append(p_idx + (dz * g + dy) * g + dx); cand_mask.append(m)
    del h_masks
    # primal p is coupled to constraint c with p = c + (3 + dz, dy, dx)  <=>  c = p - (3 + dz, dy, dx); ascending c = descending offset
    for dz, dy, dx in sorted(st15, reverse=True):
        qz, qy, qx = pz - 3 - dz, py - dy, px - dx
        ok = (qz >= 0) & (qz < g) & (qy >= 0) & (qy < g) & (qx >= 0) & (qx < g)
        c = (qz * g + qy) * g + qx
        if (dz, dy, dx) == thin:
            ok &= ~drop[np.where(ok, c, 0)]
        cand_cols.append(n1 + c); cand_mask.append(ok)
    rp1, ci1 = _from_mask(np.stack(cand_cols, axis=1), np.stack(cand_mask, axis=1))
    del cand_cols, cand_mask
    # ---- constraint rows: A columns ascending
    cand_cols, cand_mask = [], []
    for dz, dy, dx in st15:
        ok = a_valid(dz, dy, dx)
        if (dz, dy, dx) == thin:
            ok = ok & ~drop
        cand_cols.append(((cz + 3 + dz) * g + cy + dy) * g + cx + dx); cand_mask.append(ok)
    rp2, ci2 = _from_mask(np.stack(cand_cols, axis=1), np.stack(cand_mask, axis=1))
    rowptr = np.concatenate([rp1.astype(np.int64), rp1[-1].astype(np.int64) + rp2[1:].astype(np.int64)])
    assert rowptr[-1] < 2**31
    return n1 + n2, n1 + n2, rowptr.astype(np.int32), np.concatenate([ci1, ci2])


def write_mtx(path, rows, cols, rowptr, colidx, vals=None, field="real"):
    """Write a general coordinate Matrix Market file in CSR (row-major) order — through the library's threaded writer
    (``tilespmv_mtx_write``: the 4 GB text of the nlpkkt160 stand-in takes seconds, not the hours of a per-line loop)."""
    from . import api
    if field == "pattern":
        api.mtx_write(path, rows, cols, rowptr, colidx, None)
    else:
        api.mtx_write(path, rows, cols, rowptr, colidx, np.ones(len(colidx)) if vals is None else vals)
