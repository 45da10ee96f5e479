"""Parity at scale under random knob sets (round 4): the `-m gpu` fuzz uses matrices of a few thousand rows; this one takes matrices of 0.3-4 M rows from every generator family and, per
matrix, random combinations of the knobs that change HOW a plan runs — entry mode, ordered adds, column panels / column slices on XCDs (forced and timed), strip costs, split rows, descriptor
form, nontemporal streams, workgroup -> XCD maps, forced placement moves, tile-row shards — and compares the whole y with the CSR product (integer data: bit for bit), fp64 and fp32.
    python scripts/big_fuzz.py [n_matrices] [seed0]        (needs a GPU; the checker is scipy's CSR product, not the oracle: sizes the oracle would take minutes on)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scipy.sparse as sp
from tilespmv_amd import api, generators as G


def make(rng, kind):
    if kind == 0: n = int(rng.integers(300_000, 2_500_000)); return "uniform%d x %d" % (n, k := int(rng.integers(3, 12))), G.uniform_per_row(n, n + int(rng.integers(0, 40)), k, int(rng.integers(1 << 30)))
    if kind == 1: n = int(rng.integers(300_000, 2_000_000)); return "bandrand%d" % n, G.band_plus_random(n, int(rng.integers(1, 9)), int(rng.integers(1, 6)), int(rng.integers(1 << 30)))
    if kind == 2: s = int(rng.integers(18, 22)); return "rmat%d" % s, G.rmat(s, int(rng.integers(4, 12)), int(rng.integers(1 << 30)))
    if kind == 3: n = int(rng.integers(500_000, 4_000_000)); return "powerlaw%d" % n, G.powerlaw(n, seed=int(rng.integers(1 << 30)))
    if kind == 4: n = int(rng.integers(200_000, 1_000_000)); return "circuit%d" % n, G.circuit_like(n, seed=int(rng.integers(1 << 30)))
    if kind == 5: nb = int(rng.integers(2_000, 20_000)); return "blockdiag%d" % nb, G.block_diag_plus_sparse(nb, int(rng.integers(20, 70)), int(rng.integers(1, 5)), int(rng.integers(1 << 30)))
    if kind == 6: n = int(rng.integers(400, 1400)); return "lap5_%d" % n, G.laplacian5pt(n)
    if kind == 7: n = int(rng.integers(200_000, 1_500_000)); return "band%d" % n, G.band(n, int(rng.integers(2, 48)))
    # round 5: the mesh classes (FEM with 1-6 unknowns per node, shells, triangulations, tetrahedral meshes, road-like graphs), natural order or shuffled inside windows
    sh = int(rng.choice([0, 0, 16, 64, 512, 4096])); sd = int(rng.integers(1 << 30))
    if kind == 8: k = int(rng.integers(14, 40)); d = int(rng.integers(1, 7)); return "fem%d_%d s%d" % (d, k, sh), G.fem_hex(k, k + int(rng.integers(0, 5)), k, d, shuffle=sh, seed=sd)
    if kind == 9: k = int(rng.integers(150, 500)); d = int(rng.integers(1, 7)); return "shell%d_%d s%d" % (d, k, sh), G.fem_hex(k, k, 1, d, shuffle=sh, seed=sd)
    if kind == 10: k = int(rng.integers(400, 1500)); return "tri%d s%d" % (k, sh), G.tri_mesh(k, k + 3, shuffle=sh, seed=sd)
    if kind == 11: k = int(rng.integers(40, 110)); return "tet%d s%d" % (k, sh), G.tet_mesh(k, dof=int(rng.integers(1, 3)), shuffle=sh, seed=sd)
    k = int(rng.integers(600, 2000)); return "road%d s%d" % (k, sh), G.road_like(k, k, shuffle=sh, seed=sd)


def knobs(rng):
    kw = {}
    em = int(rng.integers(0, 5))
    if em < 3: kw["entry_mode"] = em
    else: kw["entry_mode"] = 2
    if rng.random() < 0.4: kw["entry_ordered"] = int(rng.integers(0, 2))
    if kw["entry_mode"] == 2:
        form = int(rng.integers(0, 5))
        if form == 1: kw.update(x_panel_kb=int(rng.choice([64, 256, 1024, 2048])), x_panel_merge=int(rng.integers(1, 5)))
        elif form == 2 and kw.get("entry_ordered") != 1: kw.update(x_panel_kb=int(rng.choice([64, 256, 1024, 2048])), x_slice_passes=int(rng.integers(1, 5)))
        elif form == 3: kw.update(x_panel_kb=int(rng.choice([128, 512, 2048])))          # the plan times the forms itself
    if rng.random() < 0.3: kw["strip_cost"] = int(rng.choice([64, 200, 800, 3200]))
    if rng.random() < 0.2: kw["split_above"] = int(rng.choice([200, 600]))
    if rng.random() < 0.3: kw["desc_dict"] = int(rng.integers(0, 2))
    if rng.random() < 0.3: kw["nt_stream"] = int(rng.integers(0, 2))
    if rng.random() < 0.3: kw.update(xcd_remap=int(rng.choice([0, 2])), xcd_chunk=int(rng.choice([8, 32, 64])))
    if rng.random() < 0.2: kw["placement_tries"] = 2
    if rng.random() < 0.2: kw["dense_mode"] = int(rng.choice([api.DENSE_MFMA, api.DENSE_VALU]))
    r = rng.random()      # round 5: what CSR-format tiles become — forced pooled / forced split on a third of the plans each way, the byte model's choice otherwise
    if r < 0.25: kw["csr_split"] = 2
    elif r < 0.4: kw["csr_split"] = 1
    elif r < 0.55: kw["csr_split"] = 3      # (second half of round 5: wide pooled units)
    if rng.random() < 0.15: kw["deterministic"] = 1
    return kw


def main():
    nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0; plans = 0
    t0 = time.time()
    for i in range(nmat):
        rng = np.random.default_rng(seed0 + i)
        name, (m, n, rp, ci) = make(rng, (seed0 + i) % 13)
        rows = (m // 16) * 16; nnz = int(rp[rows])
        for dt in (np.float64, np.float32):
            vals = rng.integers(1, 4, len(ci)).astype(dt); x = rng.integers(0, 4, n).astype(dt)
            want = sp.csr_matrix((vals[:nnz].astype(np.float64), ci[:nnz], rp[:rows + 1]), shape=(rows, n)) @ x.astype(np.float64)
            tm = api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dt)
            xd = torch.from_numpy(x).cuda()
            for j in range(4 if dt == np.float64 else 2):
                kw = knobs(rng)
                shard = rng.random() < 0.25
                cuts = [0, rows // 16] if not shard else sorted({0, int(rng.integers(1, rows // 16)), rows // 16})
                yd = torch.full((rows + 16,), -9.0, dtype=xd.dtype, device="cuda")
                if kw.get("placement_tries"): os.environ["TILESPMV_PLACEMENT_FORCE"] = "1"
                infos = []
                on_device = rng.random() < 0.4 and "placement_tries" not in kw   # (second half of round 5: the plan built by tilespmv_plan_create_from_csr — tiled matrix and streams made by kernels)
                for a, b in zip(cuts[:-1], cuts[1:]):
                    if on_device: p = api.Plan.from_csr(rows, n, nnz, rp, ci, vals, dtype=dt, tilerow_begin=a, tilerow_end=b, **kw)
                    else: p = api.Plan(tm, rows, n, nnz, tilerow_begin=a, tilerow_end=b, **kw)
                    p.spmv(xd.data_ptr(), yd.data_ptr()); p.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
                    infos.append(p.info()); p.close(); plans += 1
                os.environ.pop("TILESPMV_PLACEMENT_FORCE", None)
                y = yd.cpu().numpy()
                ok = bool(np.array_equal(y[:rows].astype(np.float64), want)) and bool((y[rows:] == -9.0).all())
                bad += not ok
                print("%-22s %s rows %8d nnz %9d  %s  %s shards %d  form %s panels %s slices %s merge %s  %s" % (name, np.dtype(dt).name, rows, nnz, "ok  " if ok else "MISMATCH", "device-built" if on_device else "host-built  ", len(cuts) - 1,
                      [q["csr_form"] for q in infos], [q["x_panels"] for q in infos], [q["x_slice_passes"] for q in infos], [q["x_panel_merge"] for q in infos], kw), flush=True)
            api.Tile_destroy(tm)
    print("BIG FUZZ: %d matrices, %d plans, %d mismatches, %.0f s" % (nmat, plans, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


main()
