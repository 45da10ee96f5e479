"""tilespmv_amd — MI355X-native TileSpMV engine (host-side Python mirror of the C ABI).

The product is the C-ABI library built from ``tilespmv_amd/csrc`` (``include/tilespmv.h``);
this package is a thin ctypes binding that keeps the reference's names and argument meaning.
"""
__version__ = "0.1.0"
