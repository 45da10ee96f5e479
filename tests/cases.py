"""Shared test matrices: name -> (rows, cols, rowptr, colidx).  All deterministic."""
import numpy as np

from tilespmv_amd import generators as G


def fnv1a64(a):
    """FNV-1a 64-bit over the raw bytes of an array (vectorised in chunks would be slower to read)."""
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def digest(a):
    """Cheap order-sensitive digest for large arrays: (sum, weighted sum) in uint64 arithmetic."""
    v = np.ascontiguousarray(a).view(np.uint8).astype(np.uint64)
    w = (np.arange(v.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return "%d:%d" % (int(v.sum()), int((v * w).sum() & np.uint64(0xFFFFFFFFFFFFFFFF)))


SMALL = {
    "lap64": lambda: G.laplacian5pt(64),
    "band4096_8": lambda: G.band(4096, 8),
    "band4096_40": lambda: G.band(4096, 40),
    "band1000_3": lambda: G.band(1000, 3),          # rows truncated to 992 by the driver rule
    "allfmt": lambda: G.all_formats(),
    "allfmt_pad5": lambda: G.all_formats(cols_pad=5),  # last tile column has 5 columns
    "rand500x700": lambda: G.random_uniform(500, 700, 0.02, 3),
    "powerlaw20k": lambda: G.powerlaw(20000),
    "circuit8k": lambda: G.circuit_like(8000),
    "empty_rows": lambda: G.from_coo(160, 160, [0, 1, 150, 150, 150], [5, 100, 3, 77, 159]),
    "one_long_row": lambda: G.from_coo(64, 6000, [3] * 3000 + [40] * 10, list(range(0, 6000, 2)) + list(range(10))),
    "wide_row_tiles": lambda: G.from_coo(32, 4096, np.repeat(np.arange(32), 400), np.tile(np.arange(0, 4000, 10), 32)),
}

MEDIUM = {
    "lap256": lambda: G.laplacian5pt(256),
    "kkt12": lambda: G.kkt_like(12),
    "powerlaw200k": lambda: G.powerlaw(200000),
    "circuit60k": lambda: G.circuit_like(60000),
}


def truncated_rows(m):
    """The reference driver drops the last rowA % 16 rows (src/main.cu:71)."""
    return (m // 16) * 16


def values_for(name, nnz, n, dtype, real=False):
    if real:
        rng = np.random.default_rng(12345)
        return rng.uniform(-1, 1, nnz).astype(dtype), rng.uniform(-1, 1, n).astype(dtype)
    return G.compat_values(nnz, dtype), G.compat_x(n, dtype)
