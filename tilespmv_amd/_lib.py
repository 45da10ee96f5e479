"""Loads the in-tree C-ABI libraries (``tilespmv_amd/lib/libtilespmv_{f64,f32}.so``).

There is deliberately no fallback: if the HIP extension has not been built the import of the
library raises, and the GPU entry points return an error / abort when no device is visible.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .tile_matrix import TileMatrixF32, TileMatrixF64

_ROOT = os.path.dirname(os.path.abspath(__file__))
_I = C.POINTER(C.c_int)
_U = C.POINTER(C.c_uint)
_CACHE = {}

INFO_NAMES = ["device_bytes", "stream_bytes", "nnz", "rows", "tiles", "coo_mode", "dense_mode", "kernel",
              "num_tasks", "num_split_rows", "fallback_nnz", "build_us", "upload_us", "entry_mode", "entry_ordered", "strip_cost",
              "wg_strips", "list_entries", "derived_units", "brick_order", "desc_bytes", "nt_stream", "retired_22", "retired_23", "placement_tries", "retired_25", "x_panels", "x_panel_merge", "scattered_entries", "x_slice_passes", "csr_form", "timed_choices_us", "device_build", "tile_create_us"]


KNOB_DEFAULT = -1
# tuning knobs of tilespmv_plan_options (include/tilespmv.h), in struct order after `autotune`
KNOB_NAMES = ["entry_mode", "entry_ordered", "strip_cost", "split_above", "split_cap", "xcd_remap", "xcd_chunk", "csr_split", "fix_inline",
              "coo_cost", "coo_heavy_min", "coo_piece", "strip_even", "wg_strips", "x_window", "x_stride1", "x_stride2", "mv_native", "mv_xcd_chunk", "lds_pad", "y_store", "desc_dict", "nt_stream", "x_panel_kb", "x_panel_merge", "placement_tries", "x_slice_passes", "deterministic", "absorb"]


class PlanOptions(C.Structure):
    """Mirror of the versioned tilespmv_plan_options: `size` first, unset knobs = KNOB_DEFAULT."""
    _fields_ = ([("size", C.c_uint), ("coo_mode", C.c_int), ("dense_mode", C.c_int), ("kernel", C.c_int),
                 ("tilerow_begin", C.c_int), ("tilerow_end", C.c_int), ("autotune", C.c_int)] +
                [(k, C.c_int) for k in KNOB_NAMES] + [("reserved", C.c_int * 1)])

    def __init__(self, coo_mode=0, dense_mode=0, kernel=0, tilerow_begin=0, tilerow_end=0, autotune=False, **knobs):
        super().__init__()
        self.size = C.sizeof(PlanOptions)
        self.coo_mode, self.dense_mode, self.kernel = coo_mode, dense_mode, kernel
        self.tilerow_begin, self.tilerow_end, self.autotune = tilerow_begin, tilerow_end, 1 if autotune else 0
        for k in KNOB_NAMES:
            setattr(self, k, KNOB_DEFAULT)
        for i in range(len(self.reserved)):
            self.reserved[i] = KNOB_DEFAULT
        for k, v in knobs.items():
            if k not in KNOB_NAMES:
                raise TypeError("unknown plan knob %r (known: %s)" % (k, ", ".join(KNOB_NAMES)))
            setattr(self, k, int(v))


def lib_path(dtype):
    suf = "f64" if np.dtype(dtype) == np.float64 else "f32"
    # TILESPMV_LIB_VARIANT: diagnostic builds made with `make VARIANT=... libs` (timing-only ablations; never the product)
    return os.path.join(_ROOT, "lib", "libtilespmv_%s%s.so" % (suf, os.environ.get("TILESPMV_LIB_VARIANT", "")))


def build(verbose=False):
    """Compile the HIP/C++ sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_ROOT, "csrc"), "-j8", "all"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building tilespmv_amd/csrc failed")
    return [lib_path(np.float64), lib_path(np.float32)]


def load(dtype=np.float64):
    dtype = np.dtype(dtype)
    if dtype in _CACHE:
        return _CACHE[dtype]
    path = lib_path(dtype)
    if not os.path.exists(path):
        raise RuntimeError("HIP extension %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C tilespmv_amd/csrc`); there is no CPU fallback" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL if False else C.RTLD_LOCAL)
    vt = C.c_double if dtype == np.float64 else C.c_float
    TM = TileMatrixF64 if dtype == np.float64 else TileMatrixF32
    VP, TP = C.POINTER(vt), C.POINTER(TM)
    assert lib.tilespmv_sizeof_value() == dtype.itemsize
    lib.Tile_create.argtypes = [TP, C.c_int, C.c_int, C.c_int, _I, _I, VP]
    lib.Tile_create.restype = None
    lib.Tile_create_ex.argtypes = [TP, C.c_int, C.c_int, C.c_int, _I, _I, VP, C.c_uint]
    lib.Tile_create_ex.restype = None
    lib.Tile_create_device.argtypes = [TP, C.c_int, C.c_int, C.c_int, _I, _I, VP, C.c_uint]
    lib.Tile_create_device.restype = C.c_int
    lib.tilespmv_plan_create_from_csr.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, _I, _I, VP, C.c_uint, C.POINTER(PlanOptions)]
    lib.tilespmv_plan_create_from_csr.restype = C.c_int
    lib.tilespmv_plan_create_from_device_csr.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(PlanOptions)]
    lib.tilespmv_plan_create_from_device_csr.restype = C.c_int
    lib.tilespmv_plan_stream_digests.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_longlong]
    lib.tilespmv_plan_stream_digests.restype = C.c_longlong
    lib.Tile_destroy.argtypes = [TP]
    lib.Tile_destroy.restype = None
    lib.tilespmv_cpu.argtypes = [TP, _I, _I, _I, C.POINTER(_U), C.POINTER(_I), C.POINTER(_I), C.c_int, C.c_int, C.c_int,
                                 _I, _I, VP, VP, VP, VP]
    lib.tilespmv_cpu.restype = None
    lib.mmio_allinone.argtypes = [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(VP), C.c_char_p]
    lib.mmio_allinone.restype = C.c_int
    lib.call_tilespmv_hip.argtypes = [C.c_char_p, TP, _I, _I, C.c_int, _U, _I, _I, C.c_int, C.c_int, C.c_int, _I, _I, VP,
                                      vt, VP, VP, VP]
    lib.call_tilespmv_hip.restype = None
    lib.call_tilespmv_hip_multi.argtypes = [C.c_char_p, TP, _I, _I, C.c_int, _U, _I, _I, C.c_int, C.c_int, C.c_int, _I, _I, VP,
                                            vt, VP, VP, VP, C.c_int, _I, C.c_int]
    lib.call_tilespmv_hip_multi.restype = C.c_int
    lib.tilespmv_plan_spmm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.tilespmv_plan_spmm.restype = C.c_int
    lib.tilespmv_plan_time_spmm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
    lib.tilespmv_plan_time_spmm.restype = C.c_double
    lib.tilespmv_plan_create.argtypes = [C.POINTER(C.c_void_p), TP, C.c_int, C.c_int, C.c_int, C.POINTER(PlanOptions)]
    lib.tilespmv_plan_create.restype = C.c_int
    lib.tilespmv_plan_options_init.argtypes = [C.POINTER(PlanOptions)]
    lib.tilespmv_plan_options_init.restype = None
    lib.tilespmv_plan_layout_digest.argtypes = [TP, C.c_int, C.c_int, C.c_int, C.POINTER(PlanOptions), C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong)]
    lib.tilespmv_plan_layout_digest.restype = C.c_int
    lib.tilespmv_plan_layout_stages.argtypes = [TP, C.c_int, C.c_int, C.c_int, C.POINTER(PlanOptions), C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong)]
    lib.tilespmv_plan_layout_stages.restype = C.c_int
    lib.tilespmv_plan_destroy.argtypes = [C.c_void_p]
    lib.tilespmv_plan_destroy.restype = None
    lib.tilespmv_plan_spmv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tilespmv_plan_spmv.restype = C.c_int
    lib.tilespmv_plan_spmv_n.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.tilespmv_plan_spmv_n.restype = C.c_int
    lib.tilespmv_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    lib.tilespmv_plan_info.restype = None
    lib.tilespmv_plan_time.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.tilespmv_plan_time.restype = C.c_double
    lib.tilespmv_plan_time_reference_style.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.tilespmv_plan_time_reference_style.restype = C.c_double
    lib.tilespmv_plan_reserve_spmm.argtypes = [C.c_void_p, C.c_int]
    lib.tilespmv_plan_reserve_spmm.restype = C.c_int
    lib.tilespmv_partition_tilerows.argtypes = [TP, C.c_int, _I]
    lib.tilespmv_partition_tilerows.restype = None
    lib.tilespmv_matrix_save.argtypes = [TP, C.c_int, C.c_int, C.c_int, C.c_char_p]
    lib.tilespmv_matrix_save.restype = C.c_int
    lib.tilespmv_matrix_load.argtypes = [TP, _I, _I, _I, C.c_char_p]
    lib.tilespmv_matrix_load.restype = C.c_int
    lib.tilespmv_csr_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _I, _I, VP, C.c_char_p]
    lib.tilespmv_csr_save.restype = C.c_int
    lib.tilespmv_csr_load.argtypes = [C.c_char_p, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(VP), C.c_char_p]
    lib.tilespmv_csr_load.restype = C.c_int
    lib.mmio_allinone_cached.argtypes = [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(VP), C.c_char_p, C.c_char_p, _I]
    lib.mmio_allinone_cached.restype = C.c_int
    lib.tilespmv_mtx_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, _I, _I, VP]
    lib.tilespmv_mtx_write.restype = C.c_int
    lib.tilespmv_reorder_rcm.argtypes = [C.c_int, _I, _I, _I]
    lib.tilespmv_reorder_rcm.restype = C.c_int
    lib.tilespmv_csr_permute.argtypes = [C.c_int, _I, _I, VP, _I, _I, _I, VP]
    lib.tilespmv_csr_permute.restype = C.c_int
    lib.tilespmv_csr_bandwidth.argtypes = [C.c_int, _I, _I]
    lib.tilespmv_csr_bandwidth.restype = C.c_longlong
    lib.tilespmv_permute_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    lib.tilespmv_permute_vector.restype = C.c_int
    lib.tilespmv_device_count.restype = C.c_int
    lib.tilespmv_plan_options_layout.restype = C.c_char_p
    lib.tilespmv_version.restype = C.c_char_p
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free.restype = None
    lib._free = libc.free
    lib._vt, lib._TM, lib._dtype = vt, TM, dtype
    _CACHE[dtype] = lib
    return lib


# every symbol include/tilespmv.h declares (checked by tests/test_abi.py)
DECLARED_SYMBOLS = ["Tile_create", "Tile_create_ex", "Tile_destroy", "tilespmv_cpu", "mmio_allinone", "call_tilespmv_hip",
                    "tilespmv_plan_create", "tilespmv_plan_destroy", "tilespmv_plan_spmv", "tilespmv_plan_info",
                    "tilespmv_plan_time", "tilespmv_partition_tilerows", "tilespmv_sizeof_value", "tilespmv_version",
                    "tilespmv_device_count", "tilespmv_matrix_save", "tilespmv_matrix_load", "tilespmv_plan_spmv_n",
                    "call_tilespmv_hip_multi", "tilespmv_plan_spmm", "tilespmv_plan_time_spmm", "tilespmv_plan_options_init", "tilespmv_plan_layout_digest",
                    "tilespmv_csr_save", "tilespmv_csr_load", "mmio_allinone_cached", "tilespmv_mtx_write",
                    "tilespmv_plan_time_reference_style", "tilespmv_plan_reserve_spmm", "tilespmv_plan_options_layout", "tilespmv_plan_layout_stages",
                    "Tile_create_device", "tilespmv_plan_create_from_csr", "tilespmv_plan_stream_digests", "tilespmv_plan_create_from_device_csr",
                    "tilespmv_reorder_rcm", "tilespmv_csr_permute", "tilespmv_csr_bandwidth", "tilespmv_permute_vector"]
