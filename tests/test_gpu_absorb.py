"""Round 6: list entries absorbed into the padding of neighbouring ELL units, and units that take their x from the previous unit instead of gathering
(tilespmv_plan_options.absorb; csrc/plan_tile_ops.h "absorbed list entries" / "DERIVED units").

The corner entries of a band / stencil — row 15 -> first column of the next block, row 0 -> last column of the previous one — sit in COO tiles of their own in the reference's
format (src/csr2tile.h:143-325 picks COO for tiles that sparse) and went to the strips' entry lists; the plan now moves the ones that fit into the padding slots of the ELL tile
next door and shifts that unit's window of x by -3 .. 3 columns.  In a band the units of a tile are consecutive diagonals: unit s is unit s - 1 one column further right, so it takes
the x its predecessor gathered, one lane up (a DPP row rotation), and only lane 15 loads.  Nothing of the result may change: whole y against the oracle bit for bit on the reference's compat data, in
both value types, host- and device-built plans with the same facts, every entry mode, split rows, shards, SpMM; real-valued data inside the stated tolerance and the same bits twice.
"""
import numpy as np
import pytest

from cases import MEDIUM, SMALL, truncated_rows, values_for

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float64): 1e-12, np.dtype(np.float32): 1e-5}


def _mats():
    from tilespmv_amd import generators as G

    def two_bands():   # a second diagonal band 160 columns to the right: ELL tiles with corner entries far from the diagonal, and a first block column whose left corners do not exist
        n = 1500
        r = np.repeat(np.arange(n), 6)
        c = (r + np.tile(np.array([-1, 0, 1, 159, 160, 161]), n))
        keep = (c >= 0) & (c < n)
        return G.from_coo(n, n, r[keep], c[keep])

    def holes():       # tridiagonal with every 7th sub-diagonal entry missing: rows of different lengths inside one ELL tile (padding in the middle of a unit's rows)
        n = 1200
        r = np.repeat(np.arange(n), 3)
        c = r + np.tile(np.array([-1, 0, 1]), n)
        keep = (c >= 0) & (c < n) & ~((np.tile(np.arange(3), n) == 0) & (r % 7 == 3))
        return G.from_coo(n, n, r[keep], c[keep])

    def ell16_full_row():   # an ELL tile of width 16 with one full row (no padding there) between two COO tiles whose entries want in: a first version dropped the full row's last entry
        ri, cj = [], []
        for r in range(16):
            cols = range(16, 32) if r == 6 else [16 + (r + j) % 16 for j in range(11)]
            ri += [r] * len(cols); cj += list(cols)
        for r in (3, 5, 6, 7, 10, 14): ri.append(r); cj.append(15)       # left neighbour, last column
        for r in (2, 6, 9, 13): ri.append(r); cj.append(32)              # right neighbour, first column
        for r in range(16, 32): ri.append(r); cj.append(r)
        return G.from_coo(32, 64, ri, cj)

    # (lap5_130 / lap7_34: grid lines two columns longer than a multiple of 16 — the off-diagonals' pieces are one-entry-per-row CSR tiles with a two-entry sparse tile beside them:
    #  CSR tiles host absorbed entries too)
    return {"ell16_full_row": ell16_full_row, "lap5_130": lambda: G.laplacian5pt(130), "lap7_34": lambda: G.laplacian7pt(34), "lap5_128": lambda: G.laplacian5pt(128), "lap5_100": lambda: G.laplacian5pt(100), "lap7_24": lambda: G.laplacian7pt(24),
            "band1": lambda: G.band(2000, 1), "band2": lambda: G.band(2000, 2), "band3_cols1003": lambda: G.band(1000, 3, ncols=1003), "band5": lambda: G.band(1500, 5),
            "two_bands": two_bands, "holes": holes, "kkt12": MEDIUM["kkt12"], "allfmt": SMALL["allfmt"], "allfmt_pad5": SMALL["allfmt_pad5"], "rand500x700": SMALL["rand500x700"]}


def _run(torch, plan, x, rowA):
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.full((rowA + 16,), 777.0, dtype=xd.dtype, device="cuda")
    plan.spmv(xd.data_ptr(), yd.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert (y[rowA:] == 777.0).all(), "wrote past the end of y"
    return y[:rowA]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_absorbed_entries_change_no_bit(dtype):
    import torch
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api
    O = CpuImpl("oracle", dtype)
    knob_sets = [dict(), dict(desc_dict=0), dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2, entry_ordered=1), dict(entry_mode=2, entry_ordered=0), dict(strip_cost=64, split_above=200),
                 dict(csr_split=0), dict(x_window=2), dict(coo_mode=api.COO_FALLBACK), dict(dense_mode=api.DENSE_VALU), dict(nt_stream=1), dict(xcd_remap=0)]
    absorbed_somewhere = derived_somewhere = 0
    for name, gen in _mats().items():
        m, n, rp, ci = gen()
        nnz, rowA = len(ci), truncated_rows(m)
        vals, x = values_for(name, nnz, n, dtype)
        hyb = name.startswith("allfmt")
        want = O.spmv(O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb), rowA, n, nnz, rp, ci, vals, x)["y"]
        tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb)
        for kw in knob_sets:
            facts = {}
            for absorb in (0, 1, 2):   # off; on, with derived units (the default); on, every unit gathering
                plan = api.Plan(tp, rowA, n, nnz, absorb=absorb, **kw)
                y = _run(torch, plan, x, rowA)
                facts[absorb] = plan.info()
                plan.close()
                assert np.array_equal(y, want), (name, kw, absorb, int(np.count_nonzero(y != want)))
            if facts[0]["csr_form"] == facts[1]["csr_form"]:   # (the form rule looks at what is left on the lists: with fewer entries a shard can keep the classic form that it would otherwise trade for pooled units)
                assert facts[1]["list_entries"] <= facts[0]["list_entries"], (name, kw)
            assert facts[2]["list_entries"] == facts[1]["list_entries"], (name, kw)
            assert facts[0]["derived_units"] == 0 and facts[2]["derived_units"] == 0, (name, kw)
            derived_somewhere += facts[1]["derived_units"]
            if name in ("lap5_128", "band1") and kw.get("csr_split") != 0 and not kw.get("split_above") and kw.get("coo_mode") != api.COO_FALLBACK:
                assert facts[1]["derived_units"] > 0, (name, kw)   # consecutive diagonals: every unit but the first of a tile takes its x from its predecessor
            if kw.get("coo_mode") == api.COO_FALLBACK:
                assert facts[1]["list_entries"] == facts[0]["list_entries"]        # (COO tiles go to the CSR fallback: nothing to absorb)
            absorbed_somewhere += max(0, facts[0]["list_entries"] - facts[1]["list_entries"])
            if name == "lap5_130" and not kw:
                assert facts[1]["list_entries"] * 20 < facts[0]["list_entries"], (facts[0]["list_entries"], facts[1]["list_entries"])   # the pieces of the off-diagonals go into the CSR tiles' units
            if name in ("lap5_128", "band1") and not kw:
                assert facts[0]["list_entries"] > 0 and facts[1]["list_entries"] == 0, (name, facts[0]["list_entries"], facts[1]["list_entries"])   # every corner entry fits
                assert facts[1]["stream_bytes"] < facts[0]["stream_bytes"]
        # the default is on; the device builder produces the same plan (same per-tile rule, plan_tile_ops.h)
        host = api.Plan(tp, rowA, n, nnz, deterministic=1)
        dev = api.Plan.from_csr(rowA, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb, deterministic=1)
        ih, idv = host.info(), dev.info()
        for k in ("list_entries", "stream_bytes", "desc_bytes", "num_tasks", "entry_mode", "csr_form"):
            assert ih[k] == idv[k], (name, k, ih[k], idv[k])
        assert np.array_equal(_run(torch, dev, x, rowA), want), (name, "device-built")
        # SpMM: the multi-vector kernels read the same shifted windows
        X = (np.arange(n * 8, dtype=np.int64) % 5).astype(dtype).reshape(n, 8)
        wcols = [O.spmv(O.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb), rowA, n, nnz, rp, ci, vals, np.ascontiguousarray(X[:, j]))["y"] for j in range(8)]
        for nv in (2, 4, 8):
            Xd = torch.from_numpy(np.ascontiguousarray(X[:, :nv])).cuda(); Yd = torch.full((rowA + 16, nv), -4.0, dtype=Xd.dtype, device="cuda")
            host.spmm(Xd.data_ptr(), Yd.data_ptr(), nv); torch.cuda.synchronize()
            Yh = Yd.cpu().numpy()
            assert (Yh[rowA:] == -4.0).all()
            for j in range(nv):
                assert np.array_equal(Yh[:rowA, j], wcols[j]), (name, "spmm", nv, j)
        host.close(); dev.close()
        # tile-row shards write their own rows only
        if rowA >= 64:
            yd = torch.full((rowA + 16,), -7.0, dtype=torch.from_numpy(x).dtype, device="cuda")
            xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
            cuts = [0, (rowA // 16) // 3, 2 * (rowA // 16) // 3 + 1, rowA // 16]
            for a, b in zip(cuts[:-1], cuts[1:]):
                p = api.Plan(tp, rowA, n, nnz, tilerow_begin=a, tilerow_end=b)
                p.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
                p.close()
            got = yd.cpu().numpy()
            assert np.array_equal(got[:rowA], want) and (got[rowA:] == -7.0).all(), (name, "shards")
        api.Tile_destroy(tp)
        # real-valued data: a row's products are added in another order — inside the tolerance, and the same bits twice
        vr, xr = values_for(name, nnz, n, dtype, real=True)
        wr = O.spmv(O.tile_create(rowA, n, nnz, rp, ci, vr, hyb=hyb), rowA, n, nnz, rp, ci, vr, xr)["y"].astype(np.float64)
        ri = np.repeat(np.arange(rowA), np.diff(rp[:rowA + 1]))
        bound = np.zeros(rowA); np.add.at(bound, ri, np.abs(vr[:int(rp[rowA])].astype(np.float64) * xr[ci[:int(rp[rowA])]].astype(np.float64)))
        tr = api.Tile_create(rowA, n, nnz, rp, ci, vr, dtype=dtype, hyb=hyb)
        p = api.Plan(tr, rowA, n, nnz, deterministic=1)
        y1 = _run(torch, p, xr, rowA); y2 = _run(torch, p, xr, rowA)
        p.close(); api.Tile_destroy(tr)
        assert np.array_equal(y1, y2), (name, "not reproducible")
        assert (np.abs(y1.astype(np.float64) - wr) <= TOL[np.dtype(dtype)] * bound + 1e-300).all(), (name, "real values")
    assert absorbed_somewhere > 0 and derived_somewhere > 0


def _random_neighbours(seed):
    """Tile-rows of ELL-like tiles (random width 1 .. 16, rows within a fifth of it, some rows full) with sparse tiles beside them whose few entries hug the shared boundary."""
    from tilespmv_amd import generators as G
    rng = np.random.default_rng(seed)
    ntr, nbc = int(rng.integers(2, 6)), int(rng.integers(6, 14))
    ri, cj = [], []
    for tr in range(ntr):
        b = 0
        while b < nbc:
            kind = rng.integers(0, 3)
            if kind == 0:   # ELL-like tile
                w = int(rng.integers(1, 17))
                for r in range(16):
                    ln = w if rng.random() < 0.4 else max(1, w - int(rng.integers(0, max(1, w // 5) + 1)))
                    cols = np.sort(rng.choice(16, size=ln, replace=False))
                    ri += [tr * 16 + r] * ln; cj += (b * 16 + cols).tolist()
            elif kind == 1:   # sparse tile near a boundary
                for _ in range(int(rng.integers(1, 9))):
                    c = int(rng.choice([0, 1, 2, 3, 12, 13, 14, 15, int(rng.integers(0, 16))]))
                    ri.append(tr * 16 + int(rng.integers(0, 16))); cj.append(b * 16 + c)
            b += 1
    ri.append(ntr * 16 - 1); cj.append(nbc * 16 - 1)
    return G.from_coo(ntr * 16, nbc * 16 - int(rng.integers(0, 9)), np.array(ri), np.minimum(np.array(cj), nbc * 16 - 10))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_absorbed_entries_random_neighbourhoods(dtype):
    """300 random arrangements of ELL tiles and sparse neighbours (the class in which scripts/big_fuzz.py found the one bug of the first version): default plans, the 12-byte
    descriptor form and the workgroup entry mode, host- and device-built, against the oracle bit for bit; a good share of the candidates must actually move."""
    import torch
    from oracle.oracle import CpuImpl
    from tilespmv_amd import api
    O = CpuImpl("oracle", dtype)
    moved = 0
    for seed in range(300):
        m, n, rp, ci = _random_neighbours(seed)
        nnz, rowA = len(ci), truncated_rows(m)
        vals, x = values_for("rand", nnz, n, dtype)
        want = O.spmv(O.tile_create(rowA, n, nnz, rp, ci, vals), rowA, n, nnz, rp, ci, vals, x)["y"]
        tp = api.Tile_create(rowA, n, nnz, rp, ci, vals, dtype=dtype)
        for kw in (dict(), dict(desc_dict=0), dict(entry_mode=2), dict(csr_split=0), dict(strip_cost=32, split_above=64)):
            plan = api.Plan(tp, rowA, n, nnz, **kw)
            y = _run(torch, plan, x, rowA)
            i1 = plan.info(); plan.close()
            assert np.array_equal(y, want), (seed, kw, int(np.count_nonzero(y != want)))
            if not kw:
                p0 = api.Plan(tp, rowA, n, nnz, absorb=0); moved += p0.info()["list_entries"] - i1["list_entries"]; p0.close()
        dev = api.Plan.from_csr(rowA, n, nnz, rp, ci, vals, dtype=dtype)
        assert np.array_equal(_run(torch, dev, x, rowA), want), (seed, "device-built")
        dev.close()
        api.Tile_destroy(tp)
    assert moved > 300
