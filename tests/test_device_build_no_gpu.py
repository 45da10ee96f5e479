"""The device preprocessing entry points (Tile_create_device, tilespmv_plan_create_from_csr) on a machine WITHOUT a GPU: they must fail loudly — there is no CPU
fallback behind them (the caller has Tile_create + tilespmv_plan_create for that) — and refuse the option sets that have no device path before touching any device."""
import ctypes as C

import numpy as np
import pytest

from tilespmv_amd import _lib, api, generators as G


def test_device_entry_points_fail_loudly_without_a_device():
    if _lib.load(np.float64).tilespmv_device_count() > 0:   # (asked here, not at collection time: the question initialises HIP in the test runner's process)
        pytest.skip("a GPU is visible: tests/test_gpu_device_build.py covers the device path")
    rows, cols, rp, ci = G.laplacian5pt(32)
    v = G.compat_values(len(ci), np.float64)
    with pytest.raises(RuntimeError):
        api.Tile_create_device(rows, cols, len(ci), rp, ci, v)
    with pytest.raises(RuntimeError):
        api.Plan.from_csr(rows, cols, len(ci), rp, ci, v)


def test_options_without_a_device_path_are_refused_before_any_device_call():
    rows, cols, rp, ci = G.laplacian5pt(32)
    for dtype in (np.float64, np.float32):
        v = G.compat_values(len(ci), dtype)
        for knobs in (dict(csr_split=0), dict(kernel=api.KERNEL_DIRECT), dict(coo_mode=api.COO_FALLBACK)):
            with pytest.raises(NotImplementedError):
                api.Plan.from_csr(rows, cols, len(ci), rp, ci, v, dtype=dtype, **knobs)
    lib = _lib.load(np.float64)
    tm = lib._TM()
    rp32, ci32, v64 = np.ascontiguousarray(rp, np.int32), np.ascontiguousarray(ci, np.int32), G.compat_values(len(ci), np.float64)
    rc = lib.Tile_create_device(C.byref(tm), rows, cols, len(ci), rp32.ctypes.data_as(_lib._I), ci32.ctypes.data_as(_lib._I), v64.ctypes.data_as(C.POINTER(C.c_double)), api.CREATE_HYB)
    assert rc in (-4, -1)   # HYB tiles are a host-only option (-4); without a device the missing device is reported first (-1)
