import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("TILESPMV_NUM_THREADS", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The C-ABI libraries and the oracle must exist; build them once if they do not."""
    from tilespmv_amd import _lib
    import numpy as np
    import subprocess
    if not (os.path.exists(_lib.lib_path(np.float64)) and os.path.exists(_lib.lib_path(np.float32))):
        _lib.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_f64.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"])
    yield
