"""The C-ABI libraries load on a machine without a GPU and export every symbol that
include/tilespmv.h declares; the GPU entry points fail loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tilespmv_amd import _lib, api, generators as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tilespmv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    return sorted(n for n in names if n.startswith(("Tile_", "tilespmv_", "call_tilespmv", "mmio_")))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_every_declared_symbol_is_exported(dtype):
    lib = _lib.load(dtype)
    decl = _declared()
    assert set(decl) == set(_lib.DECLARED_SYMBOLS), (decl, _lib.DECLARED_SYMBOLS)
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.tilespmv_sizeof_value() == np.dtype(dtype).itemsize
    assert b"gfx950" in lib.tilespmv_version()


def test_struct_layout_matches_header():
    """sizeof(Tile_matrix) seen by ctypes == the compiled reference's (when the shim is present)."""
    from oracle.oracle import available, lib_path
    from tilespmv_amd.tile_matrix import TileMatrixF64
    if not available("ref"):
        pytest.skip("oracle/_ref not built here")
    ref = C.CDLL(lib_path("ref", np.float64))
    assert ref.ref_sizeof_tile_matrix() == C.sizeof(TileMatrixF64)


def test_code_object_targets_gfx950():
    blob = open(_lib.lib_path(np.float64), "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_gpu_entry_points_fail_loudly_without_device():
    """No CPU fallback behind the GPU path: with no HIP device, plan creation reports an error."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tilespmv_amd import api, generators as G
m, n, rp, ci = G.laplacian5pt(16)
tm = api.Tile_create(m, n, len(ci), rp, ci, G.compat_values(len(ci)))
try:
    api.Plan(tm, m, n, len(ci))
except RuntimeError as e:
    print("RAISED", e); sys.exit(0)
print("NO ERROR"); sys.exit(0)
""" % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    assert "RAISED" in out and "NO ERROR" not in out, out


def test_missing_extension_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_ROOT", str(tmp_path))
    monkeypatch.setattr(_lib, "_CACHE", {})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(np.float64)


def test_cli_usage_and_silent_exit():
    """Reference CLI conventions (src/main.cu:18-22, :47)."""
    exe = os.path.join(ROOT, "tilespmv_amd", "bin", "test_f64")
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout == "Run the code by './test matrix.mtx'.\n"
    r = subprocess.run([exe, "foo.mtx"], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "-" * 32 + "!" * 8 + "-" * 36
