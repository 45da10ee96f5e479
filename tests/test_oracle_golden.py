"""The oracle (oracle/tilespmv_oracle.c) pinned against the reference.

1. Against the committed known-answer records in tests/golden/kat.json + allfmt_f64.npz, which
   tests/golden/make_golden.py produced by running the reference's own CPU headers
   (oracle/_ref).  Runs everywhere, including the GPU box.
2. Live, field by field, against oracle/_ref when those libraries are present (build container).
"""
import json
import os

import numpy as np
import pytest

from cases import SMALL, MEDIUM, fnv1a64, truncated_rows, values_for
from oracle.oracle import CpuImpl, available
from tilespmv_amd.tile_matrix import to_dict

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def _run(impl, name, gen, dtype, hyb, real=False):
    m, n, rp, ci = gen()
    nnz, rowA = len(ci), truncated_rows(m)
    vals, x = values_for(name, nnz, n, dtype, real)
    tm = impl.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb)
    d = to_dict(tm, rowA)
    s = impl.spmv(tm, rowA, n, nnz, rp, ci, vals, x)
    return d, s, rowA


@pytest.mark.parametrize("key", sorted(KAT))
def test_oracle_matches_known_answers(key):
    name, dt, variant = key.split("/")
    rec = KAT[key]
    d, s, rowA = _run(CpuImpl("oracle", np.dtype(dt)), name, SMALL[name], np.dtype(dt), variant == "hyb")
    assert rowA == rec["rows"]
    for k, v in rec["scalars"].items():
        assert d[k] == v, k
    for k, h in rec["fields"].items():
        assert fnv1a64(d[k]) == h, k
    for k, h in rec["spmv"].items():
        assert fnv1a64(s[k]) == h, k
    assert np.bincount(d["Format"], minlength=7).tolist() == rec["format_histogram"]
    assert s["rowblkblock"] == rec["rowblkblock"]
    assert s["errcount"] == 0 == rec["errcount"]  # reference self-check: tile SpMV == CSR golden, exactly
    assert float(s["y"].astype(np.float64).sum()) == rec["sum_y"]


def test_survey_known_answer_table():
    """Plain numbers of SURVEY.md §8(c) (obtained there from the unmodified reference)."""
    want = {
        "lap64": dict(tiles=1144, hist={1: 384, 2: 760}, rowblk=380, split=248, sum_y=404168, head=[9, 44, 50, 48], last=24),
        "band4096_8": dict(tiles=766, hist={0: 510, 4: 256}, rowblk=256, split=0, sum_y=1433836, head=[204, 240, 240, 286], last=165),
        "band4096_40": dict(tiles=1780, hist={0: 506, 4: 1274}, rowblk=510, split=508, sum_y=9369606, head=[1140, 962, 734, 710], last=1165),
        "band1000_3": dict(tiles=185, hist={1: 123, 2: 62}, rowblk=62, split=0, sum_y=136632, head=[14, 70, 40, 86], last=86),
    }
    O = CpuImpl("oracle", np.float64)
    for name, w in want.items():
        d, s, rowA = _run(O, name, SMALL[name], np.float64, False)
        hist = np.bincount(d["Format"], minlength=7)
        assert d["tilenum"] == w["tiles"]
        assert {i: int(c) for i, c in enumerate(hist) if c} == w["hist"]
        assert s["rowblkblock"] == w["rowblk"]
        assert int((s["blkcoostylerowidx"] >> 31).sum()) == w["split"]
        assert s["y"].sum() == w["sum_y"] and s["y"][:4].tolist() == w["head"] and s["y"][-1] == w["last"]


def test_oracle_full_dump_allfmt():
    z = np.load(os.path.join(HERE, "golden", "allfmt_f64.npz"))
    for hyb, tag in ((False, ""), (True, "hyb_")):
        d, s, rowA = _run(CpuImpl("oracle", np.float64), "allfmt", SMALL["allfmt"], np.float64, hyb)
        for k, v in d.items():
            assert np.array_equal(np.asarray(v), z[tag + k]), (tag, k)
        for k, v in s.items():
            assert np.array_equal(np.asarray(v), z[tag + "spmv_" + k]), (tag, k)
        hist = np.bincount(d["Format"], minlength=7)
        assert all(hist[f] > 0 for f in ((0, 1, 2, 3, 4, 5, 6) if hyb else (0, 1, 2, 4, 5, 6)))  # every format occurs


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("hyb", [False, True])
def test_oracle_vs_compiled_reference_live(dtype, hyb):
    kind = "ref_hyb" if hyb else "ref"
    if not available(kind, dtype):
        pytest.skip("oracle/_ref not built here (only possible where /root/reference is mounted)")
    O, R = CpuImpl("oracle", dtype), CpuImpl(kind, dtype)
    for name, gen in list(SMALL.items()) + list(MEDIUM.items()):
        for real in (False, True):
            do, so, rowA = _run(O, name, gen, dtype, hyb, real)
            dr, sr, _ = _run(R, name, gen, dtype, hyb, real)
            for k in do:
                assert np.array_equal(np.asarray(do[k]), np.asarray(dr[k])), (name, k)
            for k in ("y", "ptroffset1", "ptroffset2", "rowblkblock", "blkcoostylerowidx",
                      "blkcoostylerowidx_colstart", "blkcoostylerowidx_colstop"):
                assert np.array_equal(np.asarray(so[k]), np.asarray(sr[k])), (name, k, real)  # y is BIT-identical


def test_all_cores_variant_is_bit_identical():
    """The OpenMP tile-row-parallel CPU SpMV (bench.py's all-cores baseline) == the serial restatement, bit for bit."""
    for dtype in (np.float64, np.float32):
        O = CpuImpl("oracle", dtype)
        for name in ("allfmt", "circuit8k", "powerlaw20k", "lap64", "band4096_40"):
            for real in (False, True):
                m, n, rp, ci = SMALL[name]()
                nnz, rowA = len(ci), truncated_rows(m)
                vals, x = values_for(name, nnz, n, dtype, real)
                tm = O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=True)
                want = O.spmv(tm, rowA, n, nnz, rp, ci, vals, x)["y"]
                got, nt = O.spmv_all_cores(tm, rowA, n, x)
                assert nt >= 1 and np.array_equal(got, want), (name, dtype, real)


def test_mmio_loader_known_answers(tmp_path):
    kat = json.load(open(os.path.join(HERE, "golden", "mmio_kat.json")))
    O = CpuImpl("oracle", np.float64)
    for fn, rec in kat.items():
        r = O.mmio(os.path.join(HERE, "golden", fn))
        assert (r["rc"], r["m"], r["n"], r["nnz"], r["sym"]) == (rec["rc"], rec["m"], rec["n"], rec["nnz"], rec["sym"])
        assert fnv1a64(r["rowptr"]) == rec["rowptr"] and fnv1a64(r["colidx"]) == rec["colidx"] and fnv1a64(r["val"]) == rec["val"]
    r = O.mmio(os.path.join(HERE, "golden", "mmio_sym6.mtx"))
    assert r["rowptr"].tolist() == [0, 3, 5, 7, 8, 9, 11]            # SURVEY.md §8(c) loader-order KAT
    assert r["colidx"].tolist() == [2, 0, 3, 5, 1, 0, 4, 0, 2, 1, 5]
    assert O.mmio(str(tmp_path / "nope.mtx"))["rc"] == -1
    bad = tmp_path / "bad.mtx"; bad.write_text("hello world\n1 1 1\n")
    assert O.mmio(str(bad))["rc"] == -2
