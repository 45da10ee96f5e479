"""Driver for scripts/asan_host.sh: every small test matrix through Tile_create (both selection rules, truncated and
untruncated row counts), tilespmv_cpu, the cache round trip and the .mtx reader (good and malformed files)."""
import ctypes as C, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tilespmv_amd.tile_matrix import TileMatrixF64, to_dict
from tilespmv_amd import generators as G
import cases
lib = C.CDLL('' + os.environ.get('TILESPMV_ASAN_LIB', '/tmp/tilespmv_asan/libhost_asan.so') + '')
I = C.POINTER(C.c_int); D = C.POINTER(C.c_double); U = C.POINTER(C.c_uint)
def p(a, t): return a.ctypes.data_as(C.POINTER(t))
lib.Tile_create_ex.argtypes = [C.POINTER(TileMatrixF64), C.c_int, C.c_int, C.c_int, I, I, D, C.c_uint]
lib.tilespmv_cpu.argtypes = [C.POINTER(TileMatrixF64), I, I, I, C.POINTER(U), C.POINTER(I), C.POINTER(I), C.c_int, C.c_int, C.c_int, I, I, D, D, D, D]
lib.Tile_destroy.argtypes = [C.POINTER(TileMatrixF64)]
lib.tilespmv_matrix_save.argtypes = [C.POINTER(TileMatrixF64), C.c_int, C.c_int, C.c_int, C.c_char_p]
lib.tilespmv_matrix_load.argtypes = [C.POINTER(TileMatrixF64), I, I, I, C.c_char_p]
lib.mmio_allinone.argtypes = [I, I, I, I, C.POINTER(I), C.POINTER(I), C.POINTER(D), C.c_char_p]
libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
# round 3: the plan layout builder (host-only digest build: packed entry lists, brick order, x windows, fallback lists), the CSR cache, the writer
from tilespmv_amd._lib import PlanOptions
lib.tilespmv_plan_layout_digest.argtypes = [C.POINTER(TileMatrixF64), C.c_int, C.c_int, C.c_int, C.POINTER(PlanOptions), C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong)]
lib.mmio_allinone_cached.argtypes = [I, I, I, I, C.POINTER(I), C.POINTER(I), C.POINTER(D), C.c_char_p, C.c_char_p, I]
lib.tilespmv_mtx_write.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, I, I, D]
lib.tilespmv_csr_load.argtypes = [C.c_char_p, I, I, I, I, C.POINTER(I), C.POINTER(I), C.POINTER(D), C.c_char_p]
KNOBS = [dict(), dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2), dict(entry_mode=2, wg_strips=32), dict(coo_mode=2), dict(kernel=1),
         dict(x_window=1, strip_cost=64), dict(x_window=2, entry_mode=2, strip_cost=100), dict(strip_cost=32, split_above=100), dict(csr_split=0),
         # round 4: slab-paced lists (local part + sentinel), column-panel offsets, both with tiny slabs / panels and split rows
         
         dict(entry_mode=2, x_panel_kb=1, x_panel_merge=1), dict(entry_mode=2, x_panel_kb=8, x_panel_merge=3, strip_cost=64, split_above=200)]
names = list(cases.SMALL) + ["lap256", "kkt12"]
for name in names:
    m, n, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
    rows = (m // 16) * 16; nnz = len(ci)
    rp = np.ascontiguousarray(rp, np.int32); ci = np.ascontiguousarray(ci, np.int32)
    vals = G.compat_values(nnz); x = G.compat_x(n)
    for flags in (2, 3):
        for r_ in (rows, m):   # truncated and untruncated row counts
            nn = int(rp[r_])
            tm = TileMatrixF64()
            lib.Tile_create_ex(C.byref(tm), r_, n, nn, p(rp, C.c_int), p(ci, C.c_int), p(vals, C.c_double), flags)
            p1 = np.zeros(max(tm.tilenum, 1), np.int32); p2 = np.zeros(max(tm.tilenum, 1), np.int32)
            y = np.zeros(r_ + 16); yg = np.zeros(r_ + 16)
            nb = C.c_int(); a, b, c = U(), I(), I()
            lib.tilespmv_cpu(C.byref(tm), p(p1, C.c_int), p(p2, C.c_int), C.byref(nb), C.byref(a), C.byref(b), C.byref(c), r_, n, nn,
                             p(rp, C.c_int), p(ci, C.c_int), p(vals, C.c_double), p(x, C.c_double), p(y, C.c_double), p(yg, C.c_double))
            for q in (a, b, c): libc.free(C.cast(q, C.c_void_p))
            if r_ == rows:
                for kw in KNOBS:
                    o = PlanOptions(**kw); dg = C.c_ulonglong(0)
                    rc = lib.tilespmv_plan_layout_digest(C.byref(tm), r_, n, nn, C.byref(o), C.byref(dg), None)
                    assert rc == 0, (name, kw, rc)
            path = b"/tmp/tilespmv_asan/t.tspmv"
            assert lib.tilespmv_matrix_save(C.byref(tm), r_, n, nn, path) == 0
            t2 = TileMatrixF64(); ra, ca, za = C.c_int(), C.c_int(), C.c_int()
            assert lib.tilespmv_matrix_load(C.byref(t2), C.byref(ra), C.byref(ca), C.byref(za), path) == 0
            lib.Tile_destroy(C.byref(t2))
            # damaged cache files: every one must be rejected (-6 / -3 / -2) without touching memory it does not own
            raw = bytearray(open(path, "rb").read())
            rng = np.random.default_rng(len(raw))
            bad = [raw[:len(raw) // 2], raw[:120], raw + b"\0" * 16]
            for _ in range(40):
                q = bytearray(raw); pos = int(rng.integers(8, len(q))); q[pos] ^= 1 << int(rng.integers(0, 8)); bad.append(q)
            for q in bad:
                open("/tmp/tilespmv_asan/bad.tspmv", "wb").write(bytes(q))
                t3 = TileMatrixF64()
                rc = lib.tilespmv_matrix_load(C.byref(t3), C.byref(ra), C.byref(ca), C.byref(za), b"/tmp/tilespmv_asan/bad.tspmv")
                assert rc != 0, "a damaged cache file was accepted"
            lib.Tile_destroy(C.byref(tm))
    print(name, "ok", flush=True)
# reader on assorted files
import glob
for f in glob.glob(os.path.join(ROOT, 'tests', 'golden', '*.mtx')):
    mm, nn_, zz, ss = C.c_int(), C.c_int(), C.c_int(), C.c_int(); rpp, cii = I(), I(); vv = D()
    rc = lib.mmio_allinone(C.byref(mm), C.byref(nn_), C.byref(zz), C.byref(ss), C.byref(rpp), C.byref(cii), C.byref(vv), f.encode())
    if rc == 0:
        for q in (rpp, cii, vv): libc.free(C.cast(q, C.c_void_p))
    print(os.path.basename(f), rc, flush=True)
# stencil matrices large enough for stride detection / brick order / x windows
for gen in (lambda: G.laplacian7pt(24), lambda: G.nlpkkt_like(16, target_nnz=None), lambda: G.laplacian5pt(128)):
    m, n, rp, ci = gen(); rows = (m // 16) * 16
    rp = np.ascontiguousarray(rp, np.int32); ci = np.ascontiguousarray(ci, np.int32); vals = G.compat_values(len(ci)); nn = int(rp[rows])
    tm = TileMatrixF64()
    lib.Tile_create_ex(C.byref(tm), rows, n, nn, p(rp, C.c_int), p(ci, C.c_int), p(vals, C.c_double), 2)
    for kw in KNOBS + [dict(x_window=1, entry_mode=2, strip_cost=200), dict(x_window=2)]:
        o = PlanOptions(**kw); dg = C.c_ulonglong(0)
        assert lib.tilespmv_plan_layout_digest(C.byref(tm), rows, n, nn, C.byref(o), C.byref(dg), None) == 0, kw
    lib.Tile_destroy(C.byref(tm))
    print("layouts", m, "ok", flush=True)
# writer -> cached reader (parse + save, then load), damaged CSR caches
m, n, rp, ci = G.powerlaw(4000, seed=3)
rp = np.ascontiguousarray(rp, np.int32); ci = np.ascontiguousarray(ci, np.int32); vals = np.random.default_rng(2).standard_normal(len(ci))
assert lib.tilespmv_mtx_write(b"/tmp/tilespmv_asan/w.mtx", m, n, len(ci), p(rp, C.c_int), p(ci, C.c_int), p(vals, C.c_double)) == 0
for attempt in range(2):
    mm, nn_, zz, ss, hit = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(); rpp, cii = I(), I(); vv = D()
    rc = lib.mmio_allinone_cached(C.byref(mm), C.byref(nn_), C.byref(zz), C.byref(ss), C.byref(rpp), C.byref(cii), C.byref(vv), b"/tmp/tilespmv_asan/w.mtx", b"/tmp/tilespmv_asan/w.csr", C.byref(hit))
    assert rc == 0 and hit.value == attempt and zz.value == len(ci)
    for q in (rpp, cii, vv): libc.free(C.cast(q, C.c_void_p))
raw = bytearray(open("/tmp/tilespmv_asan/w.csr", "rb").read()); rng = np.random.default_rng(5)
for _ in range(60):
    q = bytearray(raw); pos = int(rng.integers(8, len(q))); q[pos] ^= 1 << int(rng.integers(0, 8))
    open("/tmp/tilespmv_asan/bad.csr", "wb").write(bytes(q[:len(q) - int(rng.integers(0, 3)) * 7]))
    mm, nn_, zz, ss = C.c_int(), C.c_int(), C.c_int(), C.c_int(); rpp, cii = I(), I(); vv = D()
    assert lib.tilespmv_csr_load(b"/tmp/tilespmv_asan/bad.csr", C.byref(mm), C.byref(nn_), C.byref(zz), C.byref(ss), C.byref(rpp), C.byref(cii), C.byref(vv), None) != 0
print("csr cache / writer ok", flush=True)
# truncated / malformed files
open('/tmp/tilespmv_asan/trunc.mtx', 'w').write("%%MatrixMarket matrix coordinate real general\n5 5 4\n1 1 1.0\n2 2")
open('/tmp/tilespmv_asan/short.mtx', 'w').write("%%MatrixMarket matrix coordinate real general\n5 5 4\n1 1 1.0\n")
open('/tmp/tilespmv_asan/junk.mtx', 'w').write("%%MatrixMarket matrix coordinate real general\n5 5 2\n1 x 1.0\n2 2 abc\n")
for f in ('trunc', 'short', 'junk'):
    mm, nn_, zz, ss = C.c_int(), C.c_int(), C.c_int(), C.c_int(); rpp, cii = I(), I(); vv = D()
    rc = lib.mmio_allinone(C.byref(mm), C.byref(nn_), C.byref(zz), C.byref(ss), C.byref(rpp), C.byref(cii), C.byref(vv), ('/tmp/tilespmv_asan/%s.mtx' % f).encode())
    print(f, rc, flush=True)
# round 6: reverse Cuthill-McKee + CSR permutation (host_reorder.cpp) — meshes, a graph with hubs, isolated rows, a rectangular [own | halo] block, bad permutations
lib.tilespmv_reorder_rcm.argtypes = [C.c_int, I, I, I]
lib.tilespmv_csr_permute.argtypes = [C.c_int, I, I, D, I, I, I, D]
lib.tilespmv_csr_bandwidth.argtypes = [C.c_int, I, I]; lib.tilespmv_csr_bandwidth.restype = C.c_longlong
for gen in (lambda: G.tri_mesh(40, 40, shuffle=128), lambda: G.tet_mesh(10, shuffle=64), lambda: G.powerlaw(4000, seed=2), lambda: G.laplacian5pt(3)):
    m, n, rp, ci = gen()
    k = min(m, n)
    rp = np.ascontiguousarray(rp[:k + 1], dtype=np.int32); ci = np.ascontiguousarray(ci[:int(rp[k])], dtype=np.int32)   # (columns >= k: the halo part of a rank's index space)
    v = np.arange(len(ci), dtype=np.float64)
    perm = np.zeros(max(k, 1), dtype=np.int32)
    assert lib.tilespmv_reorder_rcm(k, p(rp, C.c_int), p(ci, C.c_int), p(perm, C.c_int)) == 0 and sorted(perm[:k].tolist()) == list(range(k))
    orp, oci, ov = np.zeros(k + 1, dtype=np.int32), np.zeros(max(len(ci), 1), dtype=np.int32), np.zeros(max(len(ci), 1))
    assert lib.tilespmv_csr_permute(k, p(rp, C.c_int), p(ci, C.c_int), p(v, C.c_double), p(perm, C.c_int), p(orp, C.c_int), p(oci, C.c_int), p(ov, C.c_double)) == 0
    assert int(orp[k]) == len(ci) and lib.tilespmv_csr_bandwidth(k, p(orp, C.c_int), p(oci, C.c_int)) >= 0
    bad = perm.copy(); bad[0] = bad[-1]
    assert k < 2 or lib.tilespmv_csr_permute(k, p(rp, C.c_int), p(ci, C.c_int), p(v, C.c_double), p(bad, C.c_int), p(orp, C.c_int), p(oci, C.c_int), p(ov, C.c_double)) != 0
assert lib.tilespmv_reorder_rcm(0, None, None, None) == 0
print("reorder ok", flush=True)
print("DONE")
