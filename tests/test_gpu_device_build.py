"""Preprocessing on the device (device-side half; reference src/csr2tile.h:629-1020).

``Tile_create_device`` (hip_tile_create.hip: keys -> one stable radix sort -> run-length encoding -> per-tile selection and packing) must hand back the SAME
Tile_matrix as the host ``Tile_create`` (which tests/test_host.py pins byte for byte against the compiled reference): every scalar and every member array
is compared here, on the small all-format / ragged / empty-row cases, on the random ingredients of the fuzz generator (unsorted columns included: the extracted
matrix then needs the reference's pivot sort), on odd sizes (partial last tile-row / tile-column), fp64 and fp32."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from gpu_fuzz import random_matrix  # noqa: E402
from tilespmv_amd import api, generators as G  # noqa: E402
from tilespmv_amd.tile_matrix import to_dict  # noqa: E402

pytestmark = pytest.mark.gpu


def same_tile_matrix(rows, cols, rp, ci, dtype, cdna4=False, real=False):
    nnz = int(rp[rows])
    v = G.real_values(nnz, dtype) if real else G.compat_values(nnz, dtype)
    host = api.Tile_create(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4)
    dev = api.Tile_create_device(rows, cols, nnz, rp, ci, v, dtype=dtype, cdna4=cdna4)
    try:
        h, d = to_dict(host, rows), to_dict(dev, rows)
        bad = []
        for k in h:
            if isinstance(h[k], np.ndarray):
                if h[k].shape != d[k].shape or h[k].tobytes() != d[k].tobytes():
                    first = int(np.flatnonzero(h[k] != d[k])[0]) if h[k].shape == d[k].shape else -1
                    bad.append("%s (first difference at %d of %d)" % (k, first, h[k].size))
            elif h[k] != d[k]:
                bad.append("%s: host %d device %d" % (k, h[k], d[k]))
        return bad
    finally:
        api.Tile_destroy(host); api.Tile_destroy(dev)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", sorted(cases.SMALL) + sorted(cases.MEDIUM))
def test_device_tile_create_equals_host(name, dtype):
    rows, cols, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
    assert same_tile_matrix(rows, cols, rp, ci, dtype) == []


def test_device_tile_create_random_ingredients():
    """60 matrices from the fuzz generator's ingredients (every format, long rows, empty tile-rows, odd column counts; half of them with unsorted columns)."""
    for seed in range(60):
        rows, cols, rp, ci = random_matrix(1000 + seed)
        bad = same_tile_matrix(rows, cols, rp, ci, np.float64 if seed % 3 else np.float32, cdna4=seed % 5 == 0, real=seed % 2 == 0)
        assert bad == [], "seed %d: %s" % (1000 + seed, bad)


def test_device_tile_create_odd_shapes_and_empty():
    """Partial last tile-row and tile-column, a matrix without nonzeros, a single nonzero."""
    rng = np.random.default_rng(5)
    for rows, cols in [(1, 1), (17, 33), (100, 7), (1000, 999), (31, 5000)]:
        k = min(rows * cols // 3 + 1, 4000)
        key = np.unique(rng.integers(0, rows * cols, k))
        r, c, rp, ci = G.from_coo(rows, cols, key // cols, key % cols)
        assert same_tile_matrix(r, c, rp, ci, np.float64) == [], (rows, cols)
    rp = np.zeros(49, dtype=np.int32); ci = np.zeros(0, dtype=np.int32)
    assert same_tile_matrix(48, 48, rp, ci, np.float64) == []
    r, c, rp, ci = G.from_coo(48, 48, [47], [47])
    assert same_tile_matrix(r, c, rp, ci, np.float32) == []


def test_device_tile_create_large_classes():
    """One matrix of each large class at a size where tile-rows, sorts and scans run many workgroups: stencil, FEM (CSR tiles), power-law (hub rows), KKT."""
    for rows, cols, rp, ci in [G.laplacian5pt(700), G.fem_hex(24, 24, 24, 3), G.powerlaw(400000), G.kkt_like(40), G.rmat(17, 8, 3)]:
        assert same_tile_matrix(rows, cols, rp, ci, np.float64) == []
