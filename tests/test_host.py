"""Product host code (tilespmv_amd/csrc/host_*.cpp through the C ABI) against the oracle:
Tile_create byte-for-byte, tilespmv_cpu (schedule arrays + y), mmio_allinone; structural
invariants of SURVEY.md Appendix A; hypothesis property tests on random CSR."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from cases import SMALL, MEDIUM, truncated_rows, values_for
from oracle.oracle import CpuImpl
from tilespmv_amd import _lib, api, generators as G
from tilespmv_amd.tile_matrix import to_dict

HERE = os.path.dirname(os.path.abspath(__file__))


def _both(name, gen, dtype, hyb, real=False, rowA=None):
    m, n, rp, ci = gen()
    nnz = len(ci)
    rowA = truncated_rows(m) if rowA is None else rowA
    vals, x = values_for(name, nnz, n, dtype, real)
    O = CpuImpl("oracle", dtype)
    to = O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb)
    so = O.spmv(to, rowA, n, nnz, rp, ci, vals, x)
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb)
    sp = api.tilespmv_cpu(tp, rowA, n, nnz, rp, ci, vals, x, so["y_golden"])
    return to_dict(to, rowA), so, to_dict(tp, rowA), sp, tp


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("hyb", [False, True])
@pytest.mark.parametrize("name", sorted(SMALL) + sorted(MEDIUM))
def test_tile_create_and_cpu_path_match_oracle(name, dtype, hyb):
    gen = SMALL.get(name) or MEDIUM[name]
    for real in (False, True):
        do, so, dp, sp, tp = _both(name, gen, dtype, hyb, real)
        for k in do:
            assert np.array_equal(np.asarray(do[k]), np.asarray(dp[k])), (k, real)
        for k in sp:
            assert np.array_equal(np.asarray(sp[k]), np.asarray(so[k])), (k, real)
        api.Tile_destroy(tp)


@pytest.mark.parametrize("rowA", [187, 177, 1])
def test_partial_last_tile_row(rowA):
    """Tile_create accepts rowA % 16 != 0 (the CPU path of the reference handles it)."""
    do, so, dp, sp, tp = _both("allfmt_pad5", SMALL["allfmt_pad5"], np.float64, True, rowA=rowA)
    for k in do:
        assert np.array_equal(np.asarray(do[k]), np.asarray(dp[k])), k
    assert np.array_equal(sp["y"], so["y"]) and sp["errcount"] == 0


def test_structural_invariants():
    """offsets recorded by tilespmv_cpu equal the *_offset prefixes, etc."""
    for name in ("allfmt", "circuit8k", "powerlaw20k"):
        do, so, dp, sp, tp = _both(name, SMALL[name], np.float64, True)
        fmt = dp["Format"]
        off = {0: "csr_offset", 1: "coo_offset", 2: "ell_offset", 3: "hyb_offset", 4: "dns_offset", 5: "dnsrow_offset", 6: "dnscol_offset"}
        for f, key in off.items():
            sel = fmt == f
            assert np.array_equal(sp["ptroffset1"][sel], dp[key][:-1][sel])
        assert np.array_equal(sp["ptroffset2"][fmt == 0], dp["csrptr_offset"][:-1][fmt == 0])
        stored = np.diff(dp["blknnz"])
        assert np.array_equal(dp["blknnznnz"][:-1], stored.astype(np.uint8))
        assert dp["deferredcoo_ptr"][-1] == dp["coototal"]
        ntiles = sp["blkcoostylerowidx_colstop"] - sp["blkcoostylerowidx_colstart"]
        assert (ntiles[(sp["blkcoostylerowidx"] >> 31) == 1] <= 4).all()
        for r in range(len(dp["deferredcoo_ptr"]) - 1):
            c = dp["deferredcoo_colidx"][dp["deferredcoo_ptr"][r]:dp["deferredcoo_ptr"][r + 1]]
            assert (np.diff(c) > 0).all()


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 90), st.integers(1, 90), st.integers(0, 2 ** 31 - 1), st.floats(0.002, 0.6), st.booleans(), st.booleans())
def test_random_csr_property(m, n, seed, density, hyb, unsorted):
    rng = np.random.default_rng(seed)
    mask = rng.random((m, n)) < density
    if seed % 3 == 0:  # plant block structure so dense / dense-row / dense-col rules fire
        mask[: min(m, 16), : min(n, 16)] = True
        if m > 20: mask[16:32, :] = False; mask[17, : min(n, 16)] = True
        if n > 36: mask[: min(m, 16), 32:48] = False; mask[: min(m, 16), 35] = True
    ri, ci = np.nonzero(mask)
    rowptr = np.zeros(m + 1, np.int32); np.add.at(rowptr, ri + 1, 1); rowptr = np.cumsum(rowptr).astype(np.int32)
    ci = ci.astype(np.int32)
    if unsorted:  # columns inside a row in arbitrary order, as a symmetric .mtx produces
        for r in range(m):
            seg = ci[rowptr[r]:rowptr[r + 1]]
            ci[rowptr[r]:rowptr[r + 1]] = rng.permutation(seg)
    nnz = len(ci)
    vals, x = G.compat_values(nnz), G.compat_x(n)
    O = CpuImpl("oracle", np.float64)
    to = O.tile_create(m, n, nnz, rowptr, ci, vals, hyb=hyb)
    tp = api.Tile_create(m, n, nnz, rowptr, ci, vals, hyb=hyb)
    do, dp = to_dict(to, m), to_dict(tp, m)
    for k in do:
        assert np.array_equal(np.asarray(do[k]), np.asarray(dp[k])), k
    so = O.spmv(to, m, n, nnz, rowptr, ci, vals, x)
    sp = api.tilespmv_cpu(tp, m, n, nnz, rowptr, ci, vals, x, so["y_golden"])
    assert np.array_equal(sp["y"], so["y"])
    if not unsorted:
        assert sp["errcount"] == 0  # tile SpMV == CSR golden exactly (integer-valued data)
    api.Tile_destroy(tp)


def test_degenerate_matrices_host():
    O = CpuImpl("oracle", np.float64)
    for m, n, rp, ci in [(64, 80, np.zeros(65, np.int32), np.zeros(0, np.int32)),
                         (16, 16, np.zeros(17, np.int32), np.zeros(0, np.int32)),
                         (48, 48, np.r_[np.zeros(20), np.ones(29)].astype(np.int32), np.array([47], np.int32))]:
        vals = G.compat_values(len(ci)); x = G.compat_x(n)
        to = O.tile_create(m, n, len(ci), rp, ci, vals); tp = api.Tile_create(m, n, len(ci), rp, ci, vals)
        do, dp = to_dict(to, m), to_dict(tp, m)
        for k in do:
            assert np.array_equal(np.asarray(do[k]), np.asarray(dp[k])), k
        so = O.spmv(to, m, n, len(ci), rp, ci, vals, x)
        sp = api.tilespmv_cpu(tp, m, n, len(ci), rp, ci, vals, x, so["y_golden"])
        assert np.array_equal(sp["y"], so["y"]) and sp["rowblkblock"] == so["rowblkblock"]


def test_duplicate_entries_keep_reference_order():
    """Duplicate (i,j) are kept as separate nonzeros; the extracted rows go through the reference's
    unstable first-pivot sort, which the product restates (src/utils.h:103-137)."""
    rp = np.array([0, 6, 6, 9] + [9] * 14, dtype=np.int32)
    ci = np.array([40, 3, 40, 3, 40, 7, 100, 100, 2], dtype=np.int32)
    vals = np.arange(1, 10, dtype=np.float64)
    O = CpuImpl("oracle", np.float64)
    to = O.tile_create(16, 128, 9, rp, ci, vals)
    tp = api.Tile_create(16, 128, 9, rp, ci, vals)
    do, dp = to_dict(to, 16), to_dict(tp, 16)
    for k in do:
        assert np.array_equal(np.asarray(do[k]), np.asarray(dp[k])), k


def test_mmio_matches_oracle_and_known_answers(tmp_path):
    kat = json.load(open(os.path.join(HERE, "golden", "mmio_kat.json")))
    O = CpuImpl("oracle", np.float64)
    files = [os.path.join(HERE, "golden", f) for f in kat]
    # more header variants: pattern / integer / complex / skew / hermitian / comments / blank size line
    variants = {
        "pat.mtx": "%%MatrixMarket matrix coordinate pattern general\n% c\n%c2\n4 5 3\n1 1\n4 5\n2 3\n",
        "int.mtx": "%%MatrixMarket MATRIX Coordinate Integer Symmetric\n3 3 3\n2 1 7\n3 3 -2\n3 1 4\n",
        "cplx.mtx": "%%MatrixMarket matrix coordinate complex hermitian\n3 3 2\n2 1 1.5 -2\n3 3 4 0\n",
        "skew.mtx": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 1.5\n3 2 -4e-1\n",
        "blank.mtx": "%%MatrixMarket matrix coordinate real general\n\n 2 2 2\n1 2 3.25\n2 1 1e3\n",
    }
    for fn, text in variants.items():
        p = tmp_path / fn; p.write_text(text); files.append(str(p))
    for f in files:
        a, b = O.mmio(f), api.mmio_allinone(f)
        assert a["rc"] == b["rc"] == 0, f
        for k in ("m", "n", "nnz", "sym"):
            assert a[k] == b[k], (f, k)
        for k in ("rowptr", "colidx", "val"):
            assert np.array_equal(a[k], b[k]), (f, k)
    assert api.mmio_allinone(str(tmp_path / "missing.mtx"))["rc"] == -1
    bad = tmp_path / "bad.mtx"; bad.write_text("not a banner\n")
    assert api.mmio_allinone(str(bad))["rc"] == -2
    nosize = tmp_path / "nosize.mtx"; nosize.write_text("%%MatrixMarket matrix coordinate real general\n% only comments\n")
    assert api.mmio_allinone(str(nosize))["rc"] == -4
    for dtype in (np.float32,):
        a, b = CpuImpl("oracle", dtype).mmio(files[0]), api.mmio_allinone(files[0], dtype)
        assert np.array_equal(a["val"], b["val"]) and np.array_equal(a["colidx"], b["colidx"])


def test_mmio_parallel_chunks_and_number_formats(tmp_path):
    """A file big enough for several parser chunks (> 1 MiB each), every number spelling the fast path and its strtod
    fallback see, CRLF line ends, a symmetric file; a file whose entries span lines (token-stream fallback); an entry
    outside the declared size.  Same arrays as the oracle's fscanf-style reader, bit for bit."""
    rng = np.random.default_rng(4)
    m = n = 5000
    k = 150000
    ii, jj = rng.integers(1, m + 1, k), rng.integers(1, n + 1, k)
    vals = rng.standard_normal(k) * 10.0 ** rng.integers(-40, 40, k)
    fmts = ["%.17g", "%.6e", "%g", "%.3f", "%.25f", "%.15g", "%+.8E", "%.1f"]
    big = tmp_path / "big.mtx"
    with open(big, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% generated\n" + "%d %d %d\n" % (m, n, k))
        for t in range(k):
            f.write(("%d %d " + fmts[t % len(fmts)] + ("\r\n" if t % 7 == 0 else "\n")) % (ii[t], jj[t], vals[t]))
    assert os.path.getsize(big) > 3 * (1 << 20)
    sym = tmp_path / "sym.mtx"
    with open(sym, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n" + "%d %d %d\n" % (m, n, k))
        for t in range(k):
            f.write("  %d\t%d   %.12e  \n" % (max(ii[t], jj[t]), min(ii[t], jj[t]), vals[t]))
    span = tmp_path / "span.mtx"
    with open(span, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n" + "%d %d %d\n" % (m, n, 2000))
        for t in range(2000):
            f.write("%d\n%d %.17g " % (ii[t], jj[t], vals[t]))
    for dtype in (np.float64, np.float32):
        O = CpuImpl("oracle", dtype)
        for fpath in (big, sym, span):
            a, b = O.mmio(str(fpath)), api.mmio_allinone(str(fpath), dtype)
            assert a["rc"] == b["rc"] == 0 and a["nnz"] == b["nnz"] and a["sym"] == b["sym"], fpath
            for key in ("rowptr", "colidx", "val"):
                assert np.array_equal(a[key], b[key]), (fpath, key)
    bad = tmp_path / "range.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n4 1 2.0\n")
    assert api.mmio_allinone(str(bad))["rc"] == -4
    empty = tmp_path / "empty.mtx"; empty.write_text("")
    assert api.mmio_allinone(str(empty))["rc"] == -2


def test_config1_test_mtx_cpu_path():
    """BASELINE config 1: test.mtx, fp64, CPU path only — loader -> driver value rule -> Tile_create ->
    tilespmv_cpu; pass = errcount 0 and equality with the committed reference dump."""
    r = api.mmio_allinone(os.path.join(HERE, "golden", "test.mtx"))
    m, n, nnz = r["m"], r["n"], r["nnz"]
    vals, x = G.compat_values(nnz), G.compat_x(n)  # the driver overwrites file values (src/main.cu:68-69)
    rowA = truncated_rows(m)
    tp = api.Tile_create(rowA, n, nnz, r["rowptr"], r["colidx"], vals)
    yg = CpuImpl("oracle").csr_spmv(rowA, r["rowptr"], r["colidx"], vals, x)
    sp = api.tilespmv_cpu(tp, rowA, n, nnz, r["rowptr"], r["colidx"], vals, x, yg)
    assert sp["errcount"] == 0
    z = np.load(os.path.join(HERE, "golden", "allfmt_f64.npz"))
    d = to_dict(tp, rowA)
    for k, v in d.items():
        assert np.array_equal(np.asarray(v), z[k]), k
    assert np.array_equal(sp["y"], z["spmv_y"])


def test_matrix_cache_roundtrip(tmp_path):
    """tilespmv_matrix_save / _load (new): byte-identical Tile_matrix after a round trip; wrong value type rejected."""
    for name, hyb in (("allfmt", True), ("circuit8k", False), ("empty_rows", False)):
        m, n, rp, ci = SMALL[name]()
        nnz, rowA = len(ci), truncated_rows(m)
        for dtype in (np.float64, np.float32):
            vals, x = values_for(name, nnz, n, dtype, real=True)
            tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb)
            path = str(tmp_path / ("%s_%s.tspmv" % (name, np.dtype(dtype).name)))
            api.matrix_save(tp, rowA, n, nnz, path)
            tl, r2, c2, z2 = api.matrix_load(path, dtype)
            assert (r2, c2, z2) == (rowA, n, nnz)
            a, b = to_dict(tp, rowA), to_dict(tl, rowA)
            for k in a:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (name, k)
            other = np.float32 if dtype == np.float64 else np.float64
            with pytest.raises(OSError, match="-5"):
                api.matrix_load(path, other)
            api.Tile_destroy(tl); api.Tile_destroy(tp)
    with pytest.raises(OSError, match="-1"):
        api.matrix_load(str(tmp_path / "missing.tspmv"))
    junk = tmp_path / "junk.tspmv"; junk.write_bytes(b"not a cache file at all")
    with pytest.raises(OSError, match="-2"):
        api.matrix_load(str(junk))


def test_matrix_cache_rejects_damaged_files(tmp_path):
    """A truncated, bit-flipped or inconsistent cache must come back as an error (-6), never as a matrix: the plan
    builder and tilespmv_cpu index with its prefix arrays."""
    m, n, rp, ci = SMALL["allfmt"]()
    nnz, rowA = len(ci), truncated_rows(m)
    vals = G.compat_values(nnz)
    tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, hyb=True)
    good = tmp_path / "good.tspmv"
    api.matrix_save(tp, rowA, n, nnz, str(good))
    raw = bytearray(good.read_bytes())
    api.Tile_destroy(tp)
    header = 8 + 4 * 4 + 14 * 4 + 2 * 8 + 2 * 8
    def load_bytes(b):
        f = tmp_path / "bad.tspmv"; f.write_bytes(bytes(b))
        return api.matrix_load(str(f))
    with pytest.raises(OSError, match="-6"):
        load_bytes(raw[:-100])                                  # truncated payload
    with pytest.raises(OSError, match="-6"):
        load_bytes(raw + b"\0" * 8)                             # trailing bytes
    with pytest.raises(OSError, match="-3|-6"):
        load_bytes(raw[:header - 4])                            # truncated header
    flipped = bytearray(raw); flipped[header + 4 * (rowA // 16 + 1) + 7] ^= 0x40   # one bit in tile_columnidx
    with pytest.raises(OSError, match="-6"):
        load_bytes(flipped)
    neg = bytearray(raw); neg[8 + 16 + 2 * 4: 8 + 16 + 3 * 4] = (-5).to_bytes(4, "little", signed=True)   # tilenum < 0
    with pytest.raises(OSError, match="-6"):
        load_bytes(neg)
    big = bytearray(raw); big[8 + 16 + 3 * 4: 8 + 16 + 4 * 4] = (1 << 30).to_bytes(4, "little")             # csrsize absurd
    with pytest.raises(OSError, match="-6"):
        load_bytes(big)
    old = bytearray(raw); old[7] = ord("1")                     # the round-1 format: no checksum -> not accepted
    with pytest.raises(OSError, match="-2"):
        load_bytes(old)
    tl, r2, c2, z2 = load_bytes(raw)                            # and the intact bytes still load
    assert (r2, c2, z2) == (rowA, n, nnz)
    api.Tile_destroy(tl)


def test_csr_cache_parse_once(tmp_path):
    """mmio_allinone_cached (new): first call parses the text and writes the cache, second call reads it (same
    arrays, bit for bit, as the oracle's reader); a touched / rewritten .mtx makes the cache stale; damaged caches are refused."""
    import ctypes as C
    m, n, rp, ci = G.powerlaw(3000, seed=5)
    vals = np.random.default_rng(1).standard_normal(len(ci))
    mtx = str(tmp_path / "a.mtx"); cache = str(tmp_path / "a.csr_f64")
    api.mtx_write(mtx, m, n, rp, ci, vals)
    want = CpuImpl("oracle", np.float64).mmio(mtx)
    first = api.mmio_allinone(mtx, cache=cache)
    assert first["rc"] == 0 and first["from_cache"] == 0 and os.path.exists(cache)
    second = api.mmio_allinone(mtx, cache=cache)
    assert second["from_cache"] == 1
    for r in (first, second):
        assert (r["m"], r["n"], r["nnz"], r["sym"]) == (want["m"], want["n"], want["nnz"], want["sym"])
        for k in ("rowptr", "colidx", "val"):
            assert np.array_equal(r[k], want[k]), k
    # fp32 library refuses the fp64 cache (and re-parses into its own)
    r32 = api.mmio_allinone(mtx, np.float32, cache=cache)
    assert r32["rc"] == 0 and r32["from_cache"] == 0 and r32["val"].dtype == np.float32
    assert api.mmio_allinone(mtx, cache=cache)["from_cache"] == 0      # ... which replaced it: stale for fp64 again, rewritten
    assert api.mmio_allinone(mtx, cache=cache)["from_cache"] == 1
    # the text changes -> stale -> parsed again
    vals2 = vals.copy(); vals2[0] = 42.0
    os.utime(mtx, ns=(1, 1))
    api.mtx_write(mtx, m, n, rp, ci, vals2)
    third = api.mmio_allinone(mtx, cache=cache)
    assert third["from_cache"] == 0 and third["val"][0] == 42.0
    assert api.mmio_allinone(mtx, cache=cache)["from_cache"] == 1
    # damaged cache: flipped payload byte, truncated, junk
    lib = _lib.load(np.float64)
    blob = bytearray(open(cache, "rb").read())

    def load(path, src=None):
        mm, nn, zz, ss = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        a, b, c = _lib._I(), _lib._I(), C.POINTER(C.c_double)()
        rc = lib.tilespmv_csr_load(path.encode(), C.byref(mm), C.byref(nn), C.byref(zz), C.byref(ss), C.byref(a), C.byref(b), C.byref(c), src.encode() if src else None)
        if rc == 0:
            for q in (a, b, c):
                lib._free(C.cast(q, C.c_void_p))
        return rc
    assert load(cache) == 0 and load(cache, mtx) == 0
    bad = str(tmp_path / "bad.csr")
    flipped = bytearray(blob); flipped[len(flipped) // 2] ^= 0x40
    open(bad, "wb").write(flipped); assert load(bad) == -6
    open(bad, "wb").write(blob[:len(blob) - 9]); assert load(bad) == -6
    open(bad, "wb").write(b"not a cache at all"); assert load(bad) == -2
    assert load(str(tmp_path / "missing.csr")) == -1
    os.utime(mtx, ns=(5, 5)); assert load(cache, mtx) == -7           # stale: source mtime changed
    # an unwritable cache path does not break the load
    r = api.mmio_allinone(mtx, cache=str(tmp_path / "no_such_dir" / "x.csr"))
    assert r["rc"] == 0 and r["from_cache"] == -1


def test_mtx_writer_round_trip(tmp_path):
    """tilespmv_mtx_write: threaded writer; integers as integers, everything else %.17g (exact); pattern files."""
    rng = np.random.default_rng(9)
    m, n, rp, ci = G.random_uniform(70000, 300, 0.002, 4)    # more than one block of 65,536 rows
    vals = rng.standard_normal(len(ci)) * 10.0 ** rng.integers(-30, 30, len(ci))
    vals[::3] = rng.integers(-5000, 5000, len(vals[::3]))
    p = str(tmp_path / "w.mtx")
    api.mtx_write(p, m, n, rp, ci, vals)
    r = api.mmio_allinone(p)
    assert r["rc"] == 0 and (r["m"], r["n"], r["nnz"]) == (m, n, len(ci))
    assert np.array_equal(r["rowptr"], rp) and np.array_equal(r["colidx"], ci) and np.array_equal(r["val"], vals)
    a = CpuImpl("oracle", np.float64).mmio(p)
    assert np.array_equal(a["val"], vals) and np.array_equal(a["colidx"], ci)
    api.mtx_write(p, m, n, rp, ci, None)
    r = api.mmio_allinone(p)
    assert r["nnz"] == len(ci) and np.all(r["val"] == 1.0) and np.array_equal(r["colidx"], ci)
    G.write_mtx(p, m, n, rp, ci, vals)                        # the generators' helper goes through the same writer
    assert np.array_equal(api.mmio_allinone(p)["val"], vals)


def test_partition_tilerows_balanced():
    m, n, rp, ci = MEDIUM["powerlaw200k"]()
    nnz = len(ci); rowA = truncated_rows(m)
    tp = api.Tile_create(rowA, n, nnz, rp, ci, G.compat_values(nnz))
    d = to_dict(tp, rowA)
    for parts in (1, 2, 3, 8):
        b = api.partition_tilerows(tp, parts)
        assert b[0] == 0 and b[-1] == d["tilem"] and (np.diff(b) >= 0).all()
        load = np.diff(d["blknnz"][d["tile_ptr"][b]])
        assert load.sum() == d["blknnz"][-1]
        if parts > 1:
            assert load.max() <= 1.5 * load.sum() / parts + 4096


def test_stand_in_generators_have_the_sizes_of_the_real_matrices():
    """bench.build_matrix: the scircuit / webbase-1M stand-ins have exactly the rows and nnz of the SuiteSparse matrices
    (reference src/external/CSR5_cuda/2757-matrix.csv:544, :2379); nlpkkt_like is symmetric, column-sorted, hits a requested
    nnz exactly and has the row count of nlpkkt160 at g = 160 (8,345,600 = 2*160^3 + 6*160^2; :1903)."""
    import sys
    import scipy.sparse as sp
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        import bench
    finally:
        sys.argv = argv
    for name, rows, nnz in (("scircuit", 170998, 958936), ("webbase", 1000005, 3105536)):
        m, n, rp, ci, src = bench.build_matrix(name)
        assert (m, n, len(ci), int(rp[-1])) == (rows, rows, nnz, nnz) and "synthetic" in src
        assert (np.diff(rp) >= 0).all() and ci.min() >= 0 and ci.max() < n
        seg_sorted = np.ones(len(ci), bool); seg_sorted[1:] = np.diff(ci.astype(np.int64)) > 0
        seg_sorted[rp[:-1][np.diff(rp) > 0]] = True
        assert seg_sorted.all()                                   # columns ascending inside every row, no duplicates
    g = 12
    m, n, rp, ci = G.nlpkkt_like(g, target_nnz=None)
    assert m == n == 2 * g ** 3 + 6 * g ** 2
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(m, n))
    assert (A != A.T).nnz == 0 and A.has_sorted_indices
    full = len(ci)
    m2, n2, rp2, ci2 = G.nlpkkt_like(g, target_nnz=full - 2 * 137)
    A2 = sp.csr_matrix((np.ones(len(ci2)), ci2, rp2), shape=(m2, n2))
    assert len(ci2) == full - 2 * 137 and (A2 != A2.T).nnz == 0
    assert 2 * 160 ** 3 + 6 * 160 ** 2 == G.NLPKKT160_ROWS
    m3, n3, rp3, ci3 = G.retarget_nnz(*G.circuit_like(3000, seed=4), target_nnz=20000, seed=4)
    assert len(ci3) == 20000 and int(rp3[-1]) == 20000
