"""Host-side mirror of the reference's driver interface (same names and argument meaning as
reference src/main.cu calls them), over the C ABI of ``include/tilespmv.h``.

    tm = Tile_create(rowA, colA, nnzA, rowptr, colidx, vals)        # src/csr2tile.h:629
    sched = tilespmv_cpu(tm, rowA, colA, nnzA, rowptr, colidx, vals, x, y_golden)   # src/tilespmv_cpu.h:3
    y = call_tilespmv_hip("A.mtx", tm, sched, rowA, colA, nnzA, rowptr, colidx, vals, x)  # src/tilespmv_cuda.h:794
    plan = Plan(tm, rowA, colA, nnzA); plan.spmv(x_dev_ptr, y_dev_ptr)            # resident-plan API (new)
"""
import ctypes as C

import numpy as np

from . import _lib
from .tile_matrix import to_dict  # noqa: F401  (re-export)

_I, _U = _lib._I, _lib._U

COO_AUTO, COO_IN_TILE, COO_FALLBACK = 0, 1, 2
DENSE_AUTO, DENSE_MFMA, DENSE_VALU = 0, 1, 2
KERNEL_AUTO, KERNEL_DIRECT, KERNEL_STREAM = 0, 1, 2
CREATE_HYB, CREATE_QUIET, CREATE_CDNA4 = 1, 2, 4


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _take(lib, ptr, n, dtype):
    if not ptr:
        return np.zeros(0, dtype=dtype)
    addr = C.cast(ptr, C.c_void_p).value
    out = np.frombuffer((C.c_char * (max(n, 0) * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype, count=max(n, 0)).copy()
    lib._free(C.cast(ptr, C.c_void_p))
    return out


def _csr(lib, rowptr, colidx, vals):
    return (np.ascontiguousarray(rowptr, dtype=np.int32), np.ascontiguousarray(colidx, dtype=np.int32),
            np.ascontiguousarray(vals, dtype=lib._dtype))


def Tile_create(rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, dtype=None, hyb=False, quiet=True, cdna4=False):
    dtype = np.dtype(dtype or np.asarray(csrValA).dtype)
    lib = _lib.load(dtype)
    rp, ci, v = _csr(lib, csrRowPtrA, csrColIdxA, csrValA)
    tm = lib._TM()
    flags = (CREATE_HYB if hyb else 0) | (CREATE_QUIET if quiet else 0) | (CREATE_CDNA4 if cdna4 else 0)
    lib.Tile_create_ex(C.byref(tm), rowA, colA, nnzA, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, lib._vt), flags)
    tm._keep = (rp, ci, v)
    tm._lib = lib
    return tm


def Tile_create_device(rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, dtype=None, quiet=True, cdna4=False, hyb=False):
    """``Tile_create`` computed on the GPU (hip_tile_create.hip): the CSR arrays go up, the tiled matrix comes back, byte for byte what ``Tile_create`` builds
    (``hyb=True``: with the reference's dormant HYB rule switched on, as ``Tile_create(hyb=True)``).  No CPU fallback: raises when no device is visible (rc -1)."""
    dtype = np.dtype(dtype or np.asarray(csrValA).dtype)
    lib = _lib.load(dtype)
    rp, ci, v = _csr(lib, csrRowPtrA, csrColIdxA, csrValA)
    tm = lib._TM()
    flags = (CREATE_QUIET if quiet else 0) | (CREATE_CDNA4 if cdna4 else 0) | (CREATE_HYB if hyb else 0)
    rc = lib.Tile_create_device(C.byref(tm), rowA, colA, nnzA, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, lib._vt), flags)
    if rc != 0:
        raise RuntimeError("Tile_create_device failed (%d): no usable HIP device / extension, or offsets beyond int32" % rc)
    tm._keep = (rp, ci, v)
    tm._lib = lib
    return tm


def Tile_destroy(tm):
    tm._lib.Tile_destroy(C.byref(tm))


def tilespmv_cpu(tm, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, x, y_golden):
    lib = tm._lib
    rp, ci, v = _csr(lib, csrRowPtrA, csrColIdxA, csrValA)
    x = np.ascontiguousarray(x, dtype=lib._dtype)
    yg = np.ascontiguousarray(y_golden, dtype=lib._dtype)
    n = tm.tilenum
    p1 = np.zeros(max(n, 1), dtype=np.int32); p2 = np.zeros(max(n, 1), dtype=np.int32)
    y = np.zeros(rowA + 16, dtype=lib._dtype)
    nb = C.c_int(0); a, b, c = _U(), _I(), _I()
    lib.tilespmv_cpu(C.byref(tm), _p(p1, C.c_int), _p(p2, C.c_int), C.byref(nb), C.byref(a), C.byref(b), C.byref(c),
                     rowA, colA, nnzA, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, lib._vt), _p(x, lib._vt), _p(y, lib._vt), _p(yg, lib._vt))
    k = nb.value
    return {"y": y[:rowA].copy(), "ptroffset1": p1[:n].copy(), "ptroffset2": p2[:n].copy(), "rowblkblock": k,
            "blkcoostylerowidx": _take(lib, a, k, np.uint32), "blkcoostylerowidx_colstart": _take(lib, b, k, np.int32),
            "blkcoostylerowidx_colstop": _take(lib, c, k, np.int32),
            "errcount": int(np.count_nonzero(y[:rowA] != yg[:rowA]))}


def mmio_allinone(filename, dtype=np.float64, cache=None):
    """``cache``: path of the binary CSR cache kept beside the text (``mmio_allinone_cached``): read when it is fresh for
    ``filename`` (size and mtime), otherwise the text is parsed and the cache (re)written.  The result then carries
    ``from_cache`` (1 read, 0 parsed and saved, -1 parsed, cache not writable)."""
    lib = _lib.load(dtype)
    VP = C.POINTER(lib._vt)
    m, n, nnz, sym, hit = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(0)
    rp, ci, cv = _I(), _I(), VP()
    if cache is None:
        rc = lib.mmio_allinone(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(cv), filename.encode())
    else:
        rc = lib.mmio_allinone_cached(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(cv), filename.encode(),
                                      cache.encode(), C.byref(hit))
    if rc != 0:
        return {"rc": rc}
    out = {"rc": 0, "m": m.value, "n": n.value, "nnz": nnz.value, "sym": sym.value,
           "rowptr": _take(lib, rp, m.value + 1, np.int32), "colidx": _take(lib, ci, nnz.value, np.int32),
           "val": _take(lib, cv, nnz.value, lib._dtype)}
    if cache is not None:
        out["from_cache"] = hit.value
    return out


def mtx_write(path, rows, cols, rowptr, colidx, vals=None, dtype=np.float64):
    """General coordinate Matrix Market file in CSR order, written by the library's threaded writer (GBs in seconds);
    ``vals=None`` writes a pattern file."""
    lib = _lib.load(dtype)
    rp = np.ascontiguousarray(rowptr, dtype=np.int32); ci = np.ascontiguousarray(colidx, dtype=np.int32)
    v = None if vals is None else np.ascontiguousarray(vals, dtype=lib._dtype)
    rc = lib.tilespmv_mtx_write(path.encode(), rows, cols, len(ci), _p(rp, C.c_int), _p(ci, C.c_int), None if v is None else _p(v, lib._vt))
    if rc != 0:
        raise OSError("tilespmv_mtx_write(%s) failed: %d" % (path, rc))


Y_SHARDED, Y_ALLGATHER, Y_ALLREDUCE = 0, 1, 2


def call_tilespmv_hip(filename, tm, sched, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, x, alpha=1.0,
                      device_ids=None, y_combine_mode=Y_ALLGATHER):
    """Host pointers in, y (host) out — the reference's one-shot GPU entry (timing + results.csv included).
    With ``device_ids`` (a list, ids may repeat) the multi-device form ``call_tilespmv_hip_multi`` runs instead."""
    lib = tm._lib
    rp, ci, v = _csr(lib, csrRowPtrA, csrColIdxA, csrValA)
    x = np.ascontiguousarray(x, dtype=lib._dtype)
    y = np.zeros(rowA + 16, dtype=lib._dtype)
    yg = np.zeros(rowA + 16, dtype=lib._dtype)
    p1 = np.ascontiguousarray(sched["ptroffset1"], dtype=np.int32) if sched else np.zeros(max(tm.tilenum, 1), np.int32)
    p2 = np.ascontiguousarray(sched["ptroffset2"], dtype=np.int32) if sched else np.zeros(max(tm.tilenum, 1), np.int32)
    ri = np.ascontiguousarray(sched["blkcoostylerowidx"], dtype=np.uint32) if sched else np.zeros(1, np.uint32)
    c0 = np.ascontiguousarray(sched["blkcoostylerowidx_colstart"], dtype=np.int32) if sched else np.zeros(1, np.int32)
    c1 = np.ascontiguousarray(sched["blkcoostylerowidx_colstop"], dtype=np.int32) if sched else np.zeros(1, np.int32)
    args = (filename.encode(), C.byref(tm), _p(p1, C.c_int), _p(p2, C.c_int), int(sched["rowblkblock"]) if sched else 0,
            _p(ri, C.c_uint), _p(c0, C.c_int), _p(c1, C.c_int), rowA, colA, nnzA, _p(rp, C.c_int), _p(ci, C.c_int),
            _p(v, lib._vt), alpha, _p(x, lib._vt), _p(y, lib._vt), _p(yg, lib._vt))
    if device_ids is None:
        lib.call_tilespmv_hip(*args)
    else:
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        rc = lib.call_tilespmv_hip_multi(*args, len(ids), _p(ids, C.c_int), int(y_combine_mode))
        if rc != 0:
            raise RuntimeError("call_tilespmv_hip_multi failed (%d): see stderr" % rc)
    return y[:rowA].copy()


def matrix_save(tm, rowA, colA, nnzA, path):
    rc = tm._lib.tilespmv_matrix_save(C.byref(tm), rowA, colA, nnzA, path.encode())
    if rc != 0:
        raise OSError("tilespmv_matrix_save(%s) failed: %d" % (path, rc))


def matrix_load(path, dtype=np.float64):
    """Returns (Tile_matrix, rowA, colA, nnzA) read from a cache file written by matrix_save."""
    lib = _lib.load(dtype)
    tm = lib._TM()
    r, c, z = C.c_int(), C.c_int(), C.c_int()
    rc = lib.tilespmv_matrix_load(C.byref(tm), C.byref(r), C.byref(c), C.byref(z), path.encode())
    if rc != 0:
        raise OSError("tilespmv_matrix_load(%s) failed: %d" % (path, rc))
    tm._lib = lib
    return tm, r.value, c.value, z.value


def partition_tilerows(tm, nparts):
    b = np.zeros(nparts + 1, dtype=np.int32)
    tm._lib.tilespmv_partition_tilerows(C.byref(tm), nparts, _p(b, C.c_int))
    return b


def reorder_rcm(n, rowptr, colidx, dtype=np.float64):
    """Reverse Cuthill-McKee on the symmetrised pattern of the leading n x n block (host; tilespmv_reorder_rcm): ``perm[new] = old``."""
    lib = _lib.load(dtype)
    rp, ci = np.ascontiguousarray(rowptr, dtype=np.int32), np.ascontiguousarray(colidx, dtype=np.int32)
    perm = np.zeros(max(n, 1), dtype=np.int32)
    rc = lib.tilespmv_reorder_rcm(n, _p(rp, C.c_int), _p(ci, C.c_int), _p(perm, C.c_int))
    if rc != 0:
        raise RuntimeError("tilespmv_reorder_rcm failed (%d)" % rc)
    return perm[:n]


def csr_permute(n, rowptr, colidx, vals, perm, dtype=None):
    """``B = P A P^T`` for ``perm[new] = old`` (tilespmv_csr_permute): rows and the columns < n are renumbered, columns >= n (halo) stay, every row comes out in ascending column order."""
    dtype = np.dtype(dtype if dtype is not None else np.asarray(vals).dtype)
    lib = _lib.load(dtype)
    rp, ci, v = _csr(lib, rowptr, colidx, vals)
    pm = np.ascontiguousarray(perm, dtype=np.int32)
    nnz = int(rp[n])
    orp, oci, ov = np.zeros(n + 1, dtype=np.int32), np.zeros(max(nnz, 1), dtype=np.int32), np.zeros(max(nnz, 1), dtype=dtype)
    rc = lib.tilespmv_csr_permute(n, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, lib._vt), _p(pm, C.c_int), _p(orp, C.c_int), _p(oci, C.c_int), _p(ov, lib._vt))
    if rc != 0:
        raise ValueError("tilespmv_csr_permute: perm is not a permutation of 0 .. n - 1 (%d)" % rc)
    return orp, oci[:nnz], ov[:nnz]


def csr_bandwidth(n, rowptr, colidx, dtype=np.float64):
    lib = _lib.load(dtype)
    rp, ci = np.ascontiguousarray(rowptr, dtype=np.int32), np.ascontiguousarray(colidx, dtype=np.int32)
    return int(lib.tilespmv_csr_bandwidth(n, _p(rp, C.c_int), _p(ci, C.c_int)))


def permute_vector(d_in, d_out, d_perm, n, scatter=False, stream=0, dtype=np.float64):
    """Device vectors between the caller's numbering and a reordered plan's: gather ``out[i] = in[perm[i]]`` (into plan order) or scatter ``out[perm[i]] = in[i]`` (back)."""
    rc = _lib.load(dtype).tilespmv_permute_vector(d_in, d_out, d_perm, n, 1 if scatter else 0, stream)
    if rc != 0:
        raise RuntimeError("tilespmv_permute_vector failed (hipError %d)" % rc)


def plan_layout_digest(tm, rowA, colA, nnzA, coo_mode=COO_AUTO, dense_mode=DENSE_AUTO, kernel=0, tilerow_begin=0, tilerow_end=0, **knobs):
    """Host-only build of the plan layout (no GPU needed): returns (FNV-1a-64 digest of every stream, plan facts)."""
    opts = _lib.PlanOptions(coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, False, **knobs)
    d = C.c_ulonglong(0)
    out = (C.c_longlong * len(_lib.INFO_NAMES))()
    rc = tm._lib.tilespmv_plan_layout_digest(C.byref(tm), rowA, colA, nnzA, C.byref(opts), C.byref(d), out)
    if rc != 0:
        raise RuntimeError("tilespmv_plan_layout_digest failed (%d)" % rc)
    return d.value, {k: int(out[i]) for i, k in enumerate(_lib.INFO_NAMES)}


STAGE_NAMES = ["count", "choose", "cut", "emit", "order", "encode", "entries", "finish"]


def plan_layout_stages(tm, rowA, colA, nnzA, coo_mode=COO_AUTO, dense_mode=DENSE_AUTO, kernel=0, tilerow_begin=0, tilerow_end=0, **knobs):
    """Host-only build of the unit-stream layout, one digest per builder stage (``STAGE_NAMES``) + the plan facts."""
    opts = _lib.PlanOptions(coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, False, **knobs)
    st = (C.c_ulonglong * len(STAGE_NAMES))()
    out = (C.c_longlong * len(_lib.INFO_NAMES))()
    rc = tm._lib.tilespmv_plan_layout_stages(C.byref(tm), rowA, colA, nnzA, C.byref(opts), st, out)
    if rc != 0:
        raise RuntimeError("tilespmv_plan_layout_stages failed (%d)" % rc)
    return {k: int(st[i]) for i, k in enumerate(STAGE_NAMES)}, {k: int(out[i]) for i, k in enumerate(_lib.INFO_NAMES)}


class Plan:
    """Device-resident tiled matrix (or one tile-row shard of it)."""

    def __init__(self, tm, rowA, colA, nnzA, coo_mode=COO_AUTO, dense_mode=DENSE_AUTO, kernel=0, tilerow_begin=0, tilerow_end=0, autotune=False, **knobs):
        """``knobs``: the tuning fields of ``tilespmv_plan_options`` (``entry_mode=2, strip_cost=800, xcd_chunk=8, ...``;
        ``_lib.KNOB_NAMES``).  An unset knob falls back to its TILESPMV_* environment variable, then to the built-in default."""
        self.lib = tm._lib
        self.rowA, self.colA, self.nnzA = rowA, colA, nnzA
        opts = _lib.PlanOptions(coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, autotune, **knobs)
        h = C.c_void_p()
        rc = self.lib.tilespmv_plan_create(C.byref(h), C.byref(tm), rowA, colA, nnzA, C.byref(opts))
        if rc != 0 or not h:
            raise RuntimeError("tilespmv_plan_create failed (%d): no usable HIP device / extension" % rc)
        self.h = h

    @classmethod
    def from_csr(cls, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, dtype=None, cdna4=False, hyb=False, coo_mode=COO_AUTO, dense_mode=DENSE_AUTO, kernel=0, tilerow_begin=0, tilerow_end=0, autotune=False, **knobs):
        """``tilespmv_plan_create_from_csr``: the tiled matrix and the plan's streams are built on the device; only the CSR arrays cross the bus.
        Raises ``NotImplementedError`` for the options that have no device path (rc -4: first-generation kernel, CSR fallback, csr_split=0).  ``autotune=True``: every candidate is built from the one device-resident tiled matrix."""
        dtype = np.dtype(dtype or np.asarray(csrValA).dtype)
        lib = _lib.load(dtype)
        rp, ci, v = _csr(lib, csrRowPtrA, csrColIdxA, csrValA)
        self = cls.__new__(cls)
        self.lib = lib
        self.rowA, self.colA, self.nnzA = rowA, colA, nnzA
        opts = _lib.PlanOptions(coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, autotune, **knobs)
        h = C.c_void_p()
        rc = lib.tilespmv_plan_create_from_csr(C.byref(h), rowA, colA, nnzA, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, lib._vt), CREATE_QUIET | (CREATE_CDNA4 if cdna4 else 0) | (CREATE_HYB if hyb else 0), C.byref(opts))
        if rc == -4:
            raise NotImplementedError("tilespmv_plan_create_from_csr: these options have no device path")
        if rc != 0 or not h:
            raise RuntimeError("tilespmv_plan_create_from_csr failed (%d)" % rc)
        self.h = h
        return self

    @classmethod
    def from_device_csr(cls, rowA, colA, nnzA, d_rowptr, d_colidx, d_vals, dtype, cdna4=False, coo_mode=COO_AUTO, dense_mode=DENSE_AUTO, kernel=0, tilerow_begin=0, tilerow_end=0, **knobs):
        """``tilespmv_plan_create_from_device_csr``: like ``from_csr`` with the CSR arrays already in device memory — ``d_rowptr`` / ``d_colidx`` (int32) and ``d_vals`` are device
        ADDRESSES (e.g. ``tensor.data_ptr()`` of the crow / col / values tensors of a torch CSR tensor cast to int32); borrowed for the call."""
        dtype = np.dtype(dtype)
        lib = _lib.load(dtype)
        self = cls.__new__(cls)
        self.lib = lib
        self.rowA, self.colA, self.nnzA = rowA, colA, nnzA
        opts = _lib.PlanOptions(coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, False, **knobs)
        h = C.c_void_p()
        rc = lib.tilespmv_plan_create_from_device_csr(C.byref(h), rowA, colA, nnzA, C.c_void_p(d_rowptr), C.c_void_p(d_colidx), C.c_void_p(d_vals), CREATE_QUIET | (CREATE_CDNA4 if cdna4 else 0), C.byref(opts))
        if rc == -4:
            raise NotImplementedError("tilespmv_plan_create_from_device_csr: these options have no device path")
        if rc != 0 or not h:
            raise RuntimeError("tilespmv_plan_create_from_device_csr failed (%d)" % rc)
        self.h = h
        return self

    def spmv(self, d_x, d_y, stream=0):
        rc = self.lib.tilespmv_plan_spmv(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("tilespmv_plan_spmv: HIP error %d" % rc)

    def spmv_n(self, d_x, d_y, stream=0, count=1):
        rc = self.lib.tilespmv_plan_spmv_n(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream), count)
        if rc != 0:
            raise RuntimeError("tilespmv_plan_spmv_n: HIP error %d" % rc)

    def spmm(self, d_X, d_Y, nvec, stream=0):
        """Y[rows][nvec] = A X[cols][nvec], row-major device arrays (16-B aligned), nvec in {1, 2, 4, 8}."""
        rc = self.lib.tilespmv_plan_spmm(self.h, C.c_void_p(d_X), C.c_void_p(d_Y), nvec, C.c_void_p(stream))
        if rc == 801:
            raise NotImplementedError("tilespmv_plan_spmm: this plan uses the CSR fallback / whole-tile passes (no multi-vector kernel)")
        if rc != 0:
            raise RuntimeError("tilespmv_plan_spmm: HIP error %d" % rc)

    def time_spmm(self, d_X, d_Y, nvec, stream=0, warmup=10, reps=50):
        ms = self.lib.tilespmv_plan_time_spmm(self.h, C.c_void_p(d_X), C.c_void_p(d_Y), nvec, C.c_void_p(stream), warmup, reps)
        if ms < 0:
            raise RuntimeError("tilespmv_plan_time_spmm failed")
        return ms

    def time(self, d_x, d_y, stream=0, warmup=10, reps=50):
        ms = self.lib.tilespmv_plan_time(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream), warmup, reps)
        if ms < 0:
            raise RuntimeError("tilespmv_plan_time failed")
        return ms

    def time_reference_style(self, d_x, d_y, stream=0, reps=100):
        """Mean ms per SpMV by the reference's protocol: wall clock around one launch + synchronize (a C loop)."""
        ms = self.lib.tilespmv_plan_time_reference_style(self.h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(stream), reps)
        if ms < 0:
            raise RuntimeError("tilespmv_plan_time_reference_style failed")
        return ms

    def reserve_spmm(self, nvec):
        rc = self.lib.tilespmv_plan_reserve_spmm(self.h, nvec)
        if rc != 0:
            raise RuntimeError("tilespmv_plan_reserve_spmm: HIP error %d" % rc)

    def info(self):
        out = (C.c_longlong * len(_lib.INFO_NAMES))()
        self.lib.tilespmv_plan_info(self.h, out)
        return {k: int(out[i]) for i, k in enumerate(_lib.INFO_NAMES)}

    def stream_digests(self):
        """{member offset: (bytes, FNV-1a-64)} of every stream of the plan, read back from the device (test / audit aid)."""
        out = (C.c_ulonglong * (3 * 64))()
        n = self.lib.tilespmv_plan_stream_digests(self.h, out, 64)
        if n < 0 or n > 64:
            raise RuntimeError("tilespmv_plan_stream_digests failed (%d)" % n)
        return {int(out[3 * i]): (int(out[3 * i + 1]), int(out[3 * i + 2])) for i in range(n)}

    def close(self):
        if getattr(self, "h", None):
            self.lib.tilespmv_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def algorithmic_bytes(nnz, rows, cols, itemsize):
    """B_alg = nnz*(s_v+4) + 4*(m+1) + s_v*(n+m)."""
    return nnz * (itemsize + 4) + 4 * (rows + 1) + itemsize * (cols + rows)
