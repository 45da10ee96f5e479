"""ctypes mirror of ``Tile_matrix`` (include/tilespmv.h; reference src/format.h:3-56).

Field names, order and C types are the reference's; ``FIELD_LENGTHS`` states how many elements
each member array holds after ``Tile_create``, so that tests can view
them as numpy arrays and compare implementations field by field.
"""
import ctypes as C

import numpy as np

_I = C.POINTER(C.c_int)
_B = C.POINTER(C.c_ubyte)
_S = C.POINTER(C.c_char)


def _make(val_ctype, name):
    V = C.POINTER(val_ctype)

    class _TileMatrix(C.Structure):
        _fields_ = [
            ("tilem", C.c_int), ("tilen", C.c_int), ("tilenum", C.c_int),
            ("tile_ptr", _I), ("tile_columnidx", _I), ("tile_nnz", _I),
            ("Format", _S), ("blknnz", _I), ("blknnznnz", _B),
            ("dnsrowptr", _I), ("dnscolptr", _I), ("tilewidth", _S),
            ("csr_offset", _I), ("csrptr_offset", _I), ("coo_offset", _I), ("ell_offset", _I),
            ("hyb_offset", _I), ("hyb_coocount", _I), ("dns_offset", _I), ("dnsrow_offset", _I),
            ("dnscol_offset", _I), ("new_coocount", _I),
            ("Blockcsr_Val", V), ("Blockcsr_Ptr", _B), ("csr_compressedIdx", _B),
            ("csrsize", C.c_int), ("csrptrlen", C.c_int),
            ("Blockcoo_Val", V), ("coo_compressed_Idx", _B), ("coosize", C.c_int),
            ("Blockell_Val", V), ("ell_compressedIdx", _B), ("ellsize", C.c_int),
            ("Blockhyb_Val", V), ("hybIdx", _B),
            ("hybsize", C.c_int), ("hybellsize", C.c_int), ("hybcoosize", C.c_int),
            ("Blockdense_Val", V), ("dnssize", C.c_int),
            ("Blockdenserow_Val", V), ("denserowid", _S), ("dnsrowsize", C.c_int),
            ("Blockdensecol_Val", V), ("densecolid", _S), ("dnscolsize", C.c_int),
            ("coototal", C.c_int),
            ("deferredcoo_val", V), ("deferredcoo_colidx", _I), ("deferredcoo_ptr", _I),
        ]

    _TileMatrix.__name__ = name
    return _TileMatrix


TileMatrixF64 = _make(C.c_double, "TileMatrixF64")
TileMatrixF32 = _make(C.c_float, "TileMatrixF32")

SCALARS = ["tilem", "tilen", "tilenum", "csrsize", "csrptrlen", "coosize", "ellsize", "hybsize",
           "hybellsize", "hybcoosize", "dnssize", "dnsrowsize", "dnscolsize", "coototal"]

_NP = {"i": np.int32, "b": np.uint8, "s": np.int8}


def field_lengths(tm, rowA):
    """Element count of every member array of a created Tile_matrix."""
    n = tm.tilenum
    last = lambda p: (p[n] if n >= 0 else 0)
    return {
        "tile_ptr": tm.tilem + 1, "tile_columnidx": n, "tile_nnz": n + 1, "Format": n,
        "blknnz": n + 1, "blknnznnz": n + 1, "dnsrowptr": n + 1, "dnscolptr": n + 1, "tilewidth": n,
        "csr_offset": n + 1, "csrptr_offset": n + 1, "coo_offset": n + 1, "ell_offset": n + 1,
        "hyb_offset": n + 1, "hyb_coocount": n + 1, "dns_offset": n + 1, "dnsrow_offset": n + 1,
        "dnscol_offset": n + 1, "new_coocount": n + 1,
        "Blockcsr_Val": tm.csrsize, "Blockcsr_Ptr": tm.csrptrlen,
        "csr_compressedIdx": (tm.csrsize + 1) // 2,
        "Blockcoo_Val": tm.coosize, "coo_compressed_Idx": tm.coosize,
        "Blockell_Val": tm.ellsize, "ell_compressedIdx": (tm.ellsize + 1) // 2,
        "Blockhyb_Val": tm.hybellsize + tm.hybcoosize,
        "hybIdx": (tm.hybellsize + 1) // 2 + tm.hybcoosize,
        "Blockdense_Val": tm.dnssize,
        "Blockdenserow_Val": tm.dnsrowsize, "denserowid": last(tm.dnsrowptr),
        "Blockdensecol_Val": tm.dnscolsize, "densecolid": last(tm.dnscolptr),
        "deferredcoo_val": tm.coototal, "deferredcoo_colidx": tm.coototal, "deferredcoo_ptr": rowA + 1,
    }


def field_array(tm, name, length):
    """numpy view (copy) of member array ``name`` with ``length`` elements."""
    ptr = getattr(tm, name)
    ctype = dict(type(tm)._fields_)[name]._type_
    if ctype is C.c_char:
        dt = np.int8
    else:
        dt = np.dtype(ctype)
    if length <= 0 or not ptr:
        return np.zeros(0, dtype=dt)
    addr = C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (length * np.dtype(dt).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dt, count=length).copy()


def to_dict(tm, rowA):
    """All scalars and arrays of a Tile_matrix as a plain dict of python ints / numpy arrays."""
    out = {k: int(getattr(tm, k)) for k in SCALARS}
    for k, n in field_lengths(tm, rowA).items():
        out[k] = field_array(tm, k, int(n))
    return out
