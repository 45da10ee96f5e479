"""ctypes bindings to the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing in ``tilespmv_amd/`` does.

Two checkers share one interface (``CpuImpl``):

* ``kind="oracle"``  — ``oracle/liboracle_{f64,f32}.so``: our plain-C restatement
  (``oracle/tilespmv_oracle.c``), always available after ``make -C oracle``.
* ``kind="ref"`` / ``"ref_hyb"`` — ``oracle/_ref/libref*.so``: the reference's own CPU headers
  compiled in place from ``/root/reference/src`` by ``oracle/Makefile`` (prebuilt files travel to
  the GPU box; they cannot be rebuilt there).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from tilespmv_amd.tile_matrix import TileMatrixF32, TileMatrixF64, to_dict  # noqa: E402

_I = C.POINTER(C.c_int)
_U = C.POINTER(C.c_uint)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _take(ptr, n, dtype, free):
    """Copy n elements from a malloc'd C array into numpy and free it."""
    if n <= 0 or not ptr:
        if ptr:
            free(C.cast(ptr, C.c_void_p))
        return np.zeros(0, dtype=dtype)
    addr = C.cast(ptr, C.c_void_p).value
    out = np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr), dtype=dtype, count=n).copy()
    free(C.cast(ptr, C.c_void_p))
    return out


def lib_path(kind, dtype):
    suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
    if kind == "oracle":
        return os.path.join(_HERE, "liboracle_%s.so" % suf)
    if kind == "ref":
        return os.path.join(_HERE, "_ref", "libref_%s.so" % suf)
    if kind == "ref_hyb":
        return os.path.join(_HERE, "_ref", "libref_hyb_%s.so" % suf)
    raise ValueError(kind)


def available(kind, dtype=np.float64):
    return os.path.exists(lib_path(kind, dtype))


class CpuImpl:
    """Uniform front-end over the restatement and the compiled reference."""

    def __init__(self, kind="oracle", dtype=np.float64):
        self.kind = kind
        self.dtype = np.dtype(dtype)
        self.vt = C.c_double if self.dtype == np.float64 else C.c_float
        self.TM = TileMatrixF64 if self.dtype == np.float64 else TileMatrixF32
        path = lib_path(kind, dtype)
        if not os.path.exists(path):
            raise FileNotFoundError("%s missing — run `make -C oracle all ref ref-hyb`" % path)
        self.lib = C.CDLL(path)
        self.pre = "oracle_" if kind == "oracle" else "ref_"
        self.free = getattr(self.lib, self.pre + "free")
        self.free.argtypes = [C.c_void_p]
        self.free.restype = None
        assert getattr(self.lib, self.pre + "sizeof_value")() == self.dtype.itemsize

    # -- CSR -> tiles ------------------------------------------------------------------------
    def tile_create(self, rowA, colA, nnzA, rowptr, colidx, vals, hyb=False):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        tm = self.TM()
        if self.kind == "oracle":
            f = self.lib.oracle_Tile_create
            f.argtypes = [C.POINTER(self.TM), C.c_int, C.c_int, C.c_int, _I, _I, C.POINTER(self.vt), C.c_uint]
            f.restype = None
            f(C.byref(tm), rowA, colA, nnzA, _p(rowptr, C.c_int), _p(colidx, C.c_int), _p(vals, self.vt), 1 if hyb else 0)
        else:
            if hyb != (self.kind == "ref_hyb"):
                raise ValueError("HYB selection is a build-time property of the reference shim")
            f = self.lib.ref_Tile_create
            f.argtypes = [C.POINTER(self.TM), C.c_int, C.c_int, C.c_int, _I, _I, C.POINTER(self.vt)]
            f.restype = None
            sys.stdout.flush()
            f(C.byref(tm), rowA, colA, nnzA, _p(rowptr, C.c_int), _p(colidx, C.c_int), _p(vals, self.vt))
        tm._keep = (rowptr, colidx, vals)
        return tm

    def tile_dict(self, tm, rowA):
        return to_dict(tm, rowA)

    # -- schedule + serial tile SpMV ---------------------------------------------------------
    def spmv(self, tm, rowA, colA, nnzA, rowptr, colidx, vals, x, y_golden=None):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        x = np.ascontiguousarray(x, dtype=self.dtype)
        n = tm.tilenum
        p1 = np.zeros(max(n, 1), dtype=np.int32)
        p2 = np.zeros(max(n, 1), dtype=np.int32)
        y = np.zeros(max(rowA, 1) + 16, dtype=self.dtype)  # slack: reference zeroes 16 rows per tile-row
        yg = self.csr_spmv(rowA, rowptr, colidx, vals, x) if y_golden is None else np.ascontiguousarray(y_golden, dtype=self.dtype)
        VP = C.POINTER(self.vt)
        if self.kind == "oracle":
            sch = self.lib.oracle_schedule
            sch.argtypes = [C.POINTER(self.TM), C.POINTER(_U), C.POINTER(_I), C.POINTER(_I)]
            sch.restype = C.c_int
            a, b, c = _U(), _I(), _I()
            nblk = sch(C.byref(tm), C.byref(a), C.byref(b), C.byref(c))
            f = self.lib.oracle_tilespmv_cpu
            f.argtypes = [C.POINTER(self.TM), _I, _I, C.c_int, C.c_int, VP, VP, VP]
            f.restype = C.c_int
            err = f(C.byref(tm), _p(p1, C.c_int), _p(p2, C.c_int), rowA, colA, _p(x, self.vt), _p(y, self.vt), _p(yg, self.vt))
        else:
            f = self.lib.ref_tilespmv_cpu
            f.argtypes = [C.POINTER(self.TM), _I, _I, _I, C.POINTER(_U), C.POINTER(_I), C.POINTER(_I),
                          C.c_int, C.c_int, C.c_int, _I, _I, VP, VP, VP, VP]
            f.restype = None
            a, b, c = _U(), _I(), _I()
            nb = C.c_int(0)
            sys.stdout.flush()
            f(C.byref(tm), _p(p1, C.c_int), _p(p2, C.c_int), C.byref(nb), C.byref(a), C.byref(b), C.byref(c),
              rowA, colA, nnzA, _p(rowptr, C.c_int), _p(colidx, C.c_int), _p(vals, self.vt),
              _p(x, self.vt), _p(y, self.vt), _p(yg, self.vt))
            nblk = nb.value
            err = int(np.count_nonzero(y[:rowA] != yg[:rowA]))
        return {
            "y": y[:rowA].copy(), "y_golden": yg[:rowA].copy(), "errcount": int(err),
            "ptroffset1": p1[:n].copy(), "ptroffset2": p2[:n].copy(), "rowblkblock": int(nblk),
            "blkcoostylerowidx": _take(a, nblk, np.uint32, self.free),
            "blkcoostylerowidx_colstart": _take(b, nblk, np.int32, self.free),
            "blkcoostylerowidx_colstop": _take(c, nblk, np.int32, self.free),
        }

    def spmv_all_cores(self, tm, rowA, colA, x):
        """Tile-row-parallel (OpenMP) tile SpMV; returns (y, threads used).  Restatement only."""
        assert self.kind == "oracle"
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.zeros(max(rowA, 1) + 16, dtype=self.dtype)
        f = self.lib.oracle_tilespmv_cpu_omp
        VP = C.POINTER(self.vt)
        f.argtypes = [C.POINTER(self.TM), C.c_int, C.c_int, VP, VP]
        f.restype = C.c_int
        nt = f(C.byref(tm), rowA, colA, _p(x, self.vt), _p(y, self.vt))
        return y[:rowA].copy(), nt

    def csr_spmv(self, rowA, rowptr, colidx, vals, x):
        """Serial CSR golden, reference src/main.cu:101-110 (same loop in every checker)."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        x = np.ascontiguousarray(x, dtype=self.dtype)
        y = np.zeros(max(rowA, 1), dtype=self.dtype)
        if self.kind == "oracle":
            f = self.lib.oracle_csr_spmv
            VP = C.POINTER(self.vt)
            f.argtypes = [C.c_int, _I, _I, VP, VP, VP]
            f.restype = None
            f(rowA, _p(rowptr, C.c_int), _p(colidx, C.c_int), _p(vals, self.vt), _p(x, self.vt), _p(y, self.vt))
            return y[:rowA] if rowA else y[:0]
        return CpuImpl("oracle", self.dtype).csr_spmv(rowA, rowptr, colidx, vals, x)

    def extracted_spmv_add(self, tm, rowA, x, y):
        assert self.kind == "oracle"
        x = np.ascontiguousarray(x, dtype=self.dtype)
        f = self.lib.oracle_extracted_spmv_add
        VP = C.POINTER(self.vt)
        f.argtypes = [C.POINTER(self.TM), C.c_int, VP, VP]
        f.restype = None
        f(C.byref(tm), rowA, _p(x, self.vt), _p(y, self.vt))
        return y

    # -- .mtx reader -------------------------------------------------------------------------
    def mmio(self, filename):
        f = getattr(self.lib, self.pre + "mmio_allinone")
        VP = C.POINTER(self.vt)
        f.argtypes = [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(VP), C.c_char_p]
        f.restype = C.c_int
        m, n, nnz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rp, ci, cv = _I(), _I(), VP()
        rc = f(C.byref(m), C.byref(n), C.byref(nnz), C.byref(sym), C.byref(rp), C.byref(ci), C.byref(cv), filename.encode())
        if rc != 0:
            return {"rc": rc}
        return {"rc": 0, "m": m.value, "n": n.value, "nnz": nnz.value, "sym": sym.value,
                "rowptr": _take(rp, m.value + 1, np.int32, self.free),
                "colidx": _take(ci, nnz.value, np.int32, self.free),
                "val": _take(cv, nnz.value, self.dtype, self.free)}
