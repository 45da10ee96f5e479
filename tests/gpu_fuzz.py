"""Randomised GPU parity fuzz (manual tool, uses the oracle): python tests/gpu_fuzz.py [seeds] [first_seed]
Every seed builds a matrix from random ingredients (every fourth seed: a 3-D stencil plus random extras, so that the brick order and
the x windows of round 3 engage) (blocks of every tile format, bands, long rows, empty
tile-rows, single entries, duplicate-free scatter), odd column counts, then checks SpMV (both kernels, both COO
modes, both dense modes, HYB on/off, with tiny strip / split thresholds) and SpMM (2/4/8) bit-exactly against
the oracle's CSR golden on small-integer data, fp64 and fp32."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.oracle import CpuImpl  # noqa: E402
from tilespmv_amd import api, generators as G  # noqa: E402


def random_matrix(seed):
    rng = np.random.default_rng(seed)
    tm = int(rng.integers(1, 40)); tn = int(rng.integers(1, 60))
    rows = 16 * tm
    cols = 16 * tn - int(rng.integers(0, 16)) if rng.random() < 0.5 else 16 * tn
    cols = max(cols, 1)
    R, Cc = [], []
    def add(r, c):
        r = np.asarray(r).ravel(); c = np.asarray(c).ravel()
        k = (r >= 0) & (r < rows) & (c >= 0) & (c < cols)
        R.append(r[k]); Cc.append(c[k])
    for _ in range(int(rng.integers(1, 4 * tm + 2))):
        kind = rng.integers(0, 9)
        br, bc = 16 * int(rng.integers(0, tm)), 16 * int(rng.integers(0, tn))
        if kind == 0:      # dense block
            rr, cc = np.meshgrid(np.arange(16), np.arange(16), indexing="ij"); keep = rng.random((16, 16)) < rng.uniform(0.75, 1.0)
            add(br + rr[keep], bc + cc[keep])
        elif kind == 1:    # full rows
            for q in rng.choice(16, int(rng.integers(1, 5)), replace=False): add(np.full(16, br + q), bc + np.arange(16))
        elif kind == 2:    # full columns
            for q in rng.choice(16, int(rng.integers(1, 5)), replace=False): add(br + np.arange(16), np.full(16, bc + q))
        elif kind == 3:    # uniform width (ELL)
            w = int(rng.integers(1, 8))
            for q in range(16): add(np.full(w, br + q), bc + rng.choice(16, w, replace=False))
        elif kind == 4:    # ragged (CSR / HYB)
            for q in range(16):
                w = int(rng.integers(0, 14)); add(np.full(w, br + q), bc + rng.choice(16, w, replace=False))
        elif kind == 5:    # a few entries (COO)
            k = int(rng.integers(1, 12)); p = rng.choice(256, k, replace=False); add(br + p // 16, bc + p % 16)
        elif kind == 6:    # band segment
            hb = int(rng.integers(1, 30)); r0 = int(rng.integers(0, rows)); n = int(rng.integers(1, 200))
            for r in range(r0, min(rows, r0 + n)): add(np.full(2 * hb + 1, r), np.arange(r - hb, r + hb + 1))
        elif kind == 7:    # one long row
            r = int(rng.integers(0, rows)); k = int(rng.integers(1, cols + 1)); add(np.full(k, r), rng.choice(cols, k, replace=False))
        else:              # scattered singles
            k = int(rng.integers(1, 300)); add(rng.integers(0, rows, k), rng.integers(0, cols, k))
    r = np.concatenate(R); c = np.concatenate(Cc)
    key = np.unique(r.astype(np.int64) * cols + c)          # no duplicates (uchar per-tile counters, SURVEY S8c hazards)
    r, c = key // cols, key % cols
    if rng.random() < 0.5:                                   # unsorted columns within rows, like a symmetric .mtx
        perm = rng.permutation(len(r)); r, c = r[perm], c[perm]
    return G.from_coo(rows, cols, r, c)


def stencil_matrix(seed):
    """Round 3: a 3-D stencil (7- or 27-point on g x g x gz cells, g a multiple of 16 so that grid lines are whole tile-rows) plus
    random scatter, a few dense blocks and long rows: grid strides exist, so the brick task order and the LDS x windows engage."""
    rng = np.random.default_rng(seed)
    g = 16 * int(rng.integers(1, 4)); gy = int(rng.integers(4, 11)); gz = int(rng.integers(4, 11))
    N = g * gy * gz
    idx = np.arange(N, dtype=np.int64); k, j, i = idx // (g * gy), (idx // g) % gy, idx % g
    R, Cc = [], []
    full = rng.random() < 0.5
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if not full and abs(dz) + abs(dy) + abs(dx) > 1: continue
                ok = (k + dz >= 0) & (k + dz < gz) & (j + dy >= 0) & (j + dy < gy) & (i + dx >= 0) & (i + dx < g)
                R.append(idx[ok]); Cc.append(idx[ok] + (dz * gy + dy) * g + dx)
    ns = int(rng.integers(0, 400)); R.append(rng.integers(0, N, ns)); Cc.append(rng.integers(0, N, ns))
    for _ in range(int(rng.integers(0, 4))):
        br, bc = 16 * int(rng.integers(0, N // 16)), 16 * int(rng.integers(0, N // 16))
        rr, cc = np.meshgrid(np.arange(16), np.arange(16), indexing="ij"); R.append((br + rr).ravel()); Cc.append((bc + cc).ravel())
    if rng.random() < 0.3:
        r = int(rng.integers(0, N)); kk = int(rng.integers(1, N)); R.append(np.full(kk, r)); Cc.append(rng.choice(N, kk, replace=False))
    r = np.concatenate(R); c = np.concatenate(Cc)
    key = np.unique(r.astype(np.int64) * N + c)
    return G.from_coo(N, N, key // N, key % N)


def check(seed):
    m, n, rp, ci = stencil_matrix(seed) if seed % 4 == 1 else random_matrix(seed)
    nnz = len(ci)
    bad = 0
    rng = np.random.default_rng(seed + 7)
    env = {}
    if seed % 3 == 0: env["TILESPMV_SPLIT_ABOVE"] = "150"
    if seed % 4 == 0: env["TILESPMV_STRIP_COST"] = "64"
    if seed % 5 == 0: env["TILESPMV_COO_HEAVY_MIN"] = "4"
    # how the COO entry lists run (round 2): default choice / per strip / per wavefront / per workgroup ordered, unordered
    env.update([{}, {"TILESPMV_WAVE_COO": "0"}, {"TILESPMV_WAVE_COO": "1"}, {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "1"},
                {"TILESPMV_WAVE_COO": "2", "TILESPMV_COO_ORDERED": "0"}][seed % 5 if seed % 2 else (seed // 2) % 5])
    if seed % 11 == 0: env["TILESPMV_STRIP_COST"] = "1600"
    # round 3: brick task order on the stencil seeds, 512-thread workgroups, resident-workgroup cap
    if seed % 4 == 1: env["TILESPMV_X_WINDOW"] = str([2, 0, 2, -1][(seed // 4) % 4])
    if seed % 6 == 3 and env.get("TILESPMV_WAVE_COO") == "2": env["TILESPMV_WG_STRIPS"] = "32"
    if seed % 9 == 5: env["TILESPMV_LDS_PAD"] = "8192"
    if seed % 4 == 2: env["TILESPMV_NT_STREAM"] = "1"   # nontemporal value / entry-record loads (by rule only on launches above 400 MB)
    if seed % 3 == 1: env["TILESPMV_DESC_DICT"] = "0"   # 12-B unit descriptors (the default takes the 4-B dictionary form wherever the patterns are few)
    # round 5: what CSR-format tiles become — pooled units on three seeds of ten (with the COO tiles and HYB remainders of their tile-rows; every unit of the plan then has the
    # pooled form), the ELL-style split on another two, the byte model's choice otherwise
    if seed % 10 in (2, 5, 8): env["TILESPMV_CSR_SPLIT"] = "2"
    elif seed % 10 in (3, 7): env["TILESPMV_CSR_SPLIT"] = "1"
    # round 4: column panels of the entry lists (panel 0 in the unit kernel, one y += launch per further panel)
    if env.get("TILESPMV_WAVE_COO") == "2" and env.get("TILESPMV_WG_STRIPS") != "32" and seed % 3 == 2:
        env.update({"TILESPMV_X_PANEL_KB": str([1, 4, 2, 16][(seed // 3) % 4]), "TILESPMV_X_PANEL_MERGE": str(1 + (seed // 7) % 3)})
        # ... or, on every other such seed with unordered adds allowed, the same lists by column slices pinned to XCDs (1-4 passes; atomic adds of the touched rows)
        if (seed // 3) % 2 == 1 and env.get("TILESPMV_COO_ORDERED") != "1":
            del env["TILESPMV_X_PANEL_MERGE"]; env["TILESPMV_X_SLICE_PASSES"] = str(1 + (seed // 5) % 4)
    os.environ.update(env)
    for dt in (np.float64, np.float32):
        vals = rng.integers(1, 4, nnz).astype(dt)
        X = rng.integers(0, 4, (n, 8)).astype(dt)
        O = CpuImpl("oracle", dt)
        gold = [O.csr_spmv(m, rp, ci, vals, np.ascontiguousarray(X[:, j])) for j in range(8)]
        tdt = torch.float64 if dt == np.float64 else torch.float32
        for hyb in (False, True):
            tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dt, hyb=hyb)
            to = O.tile_create(m, n, nnz, rp, ci, vals, hyb=hyb)
            da, db = api.to_dict(tp, m), O.tile_dict(to, m)
            for k in da:
                if not np.array_equal(np.asarray(da[k]), np.asarray(db[k])):
                    print("  seed %d: Tile_matrix field %s differs (hyb=%d %s)" % (seed, k, hyb, np.dtype(dt).name)); bad += 1
            xd = torch.from_numpy(np.ascontiguousarray(X[:, 0])).cuda()
            for kern in (1, 2):
                for coo in (1, 2):
                    for dns in (1, 2):
                        plan = api.Plan(tp, m, n, nnz, coo_mode=coo, dense_mode=dns, kernel=kern)
                        yd = torch.full((m + 16,), -9.0, dtype=tdt, device="cuda")
                        plan.spmv(xd.data_ptr(), yd.data_ptr()); plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
                        y = yd.cpu().numpy()
                        if not np.array_equal(y[:m], gold[0]) or not (y[m:] == -9.0).all():
                            print("  seed %d: SpMV mismatch kern=%d coo=%d dns=%d hyb=%d %s env=%s" % (seed, kern, coo, dns, hyb, np.dtype(dt).name, env)); bad += 1
                        if kern == 2 and coo == 1:
                            for nv in (2, 4, 8):
                                Xd = torch.from_numpy(np.ascontiguousarray(X[:, :nv])).cuda()
                                Yd = torch.full((m + 16, nv), -9.0, dtype=tdt, device="cuda")
                                try:
                                    plan.spmm(Xd.data_ptr(), Yd.data_ptr(), nv); torch.cuda.synchronize()
                                except NotImplementedError:   # plans with whole-tile passes (TILESPMV_CSR_SPLIT=0) have no SpMM
                                    break
                                Y = Yd.cpu().numpy()
                                if any(not np.array_equal(Y[:m, j], gold[j]) for j in range(nv)) or not (Y[m:] == -9.0).all():
                                    print("  seed %d: SpMM mismatch nvec=%d dns=%d hyb=%d %s env=%s" % (seed, nv, dns, hyb, np.dtype(dt).name, env)); bad += 1
                        plan.close()
            api.Tile_destroy(tp)
    # real-valued data, same plan knobs: |y - y_ref| <= tol * sum_j |a_ij x_j| (1e-12 fp64 / 1e-5 fp32)
    ri = np.repeat(np.arange(m), np.diff(rp[:m + 1]))
    for dt, tol in ((np.float64, 1e-12), (np.float32, 1e-5)):
        vals = rng.uniform(-1, 1, nnz).astype(dt); xr = rng.uniform(-1, 1, n).astype(dt)
        ref = np.zeros(m); np.add.at(ref, ri, vals[:rp[m]].astype(np.float64) * xr[ci[:rp[m]]].astype(np.float64))
        bound = np.zeros(m); np.add.at(bound, ri, np.abs(vals[:rp[m]].astype(np.float64) * xr[ci[:rp[m]]].astype(np.float64)))
        tp = api.Tile_create(m, n, nnz, rp, ci, vals, dtype=dt, hyb=bool(seed & 1))
        xd = torch.from_numpy(xr).cuda()
        for coo in (1, 2):
            plan = api.Plan(tp, m, n, nnz, coo_mode=coo)
            yd = torch.zeros(m + 16, dtype=xd.dtype, device="cuda")
            plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
            if not (np.abs(yd.cpu().numpy()[:m].astype(np.float64) - ref) <= tol * bound + 1e-300).all():
                print("  seed %d: real-valued SpMV outside tolerance coo=%d %s env=%s" % (seed, coo, np.dtype(dt).name, env)); bad += 1
            plan.close()
        api.Tile_destroy(tp)
    for k in env: os.environ.pop(k)
    return bad, (m, n, nnz)


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    total = 0
    for s in range(first, first + seeds):
        b, shape = check(s)
        total += b
        print("seed %d %s -> %d mismatches" % (s, shape, b), flush=True)
    print("TOTAL MISMATCHES", total, flush=True)
    sys.exit(1 if total else 0)
