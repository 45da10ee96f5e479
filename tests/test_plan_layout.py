"""Plan layout built on the host only (tilespmv_plan_layout_digest: the exact builder of tilespmv_plan_create, no HIP call):
versioned options, knobs through the struct instead of the environment, packed entry lists that decode back to their entries
(checked inside the builder), two threads building differently tuned layouts.  No GPU needed."""
import ctypes as C
import threading

import numpy as np
import pytest

from tests import cases
from tilespmv_amd import _lib, api, generators as G


def _tm(name, dtype=np.float64, hyb=False):
    m, n, rp, ci = (cases.SMALL.get(name) or cases.MEDIUM[name])()
    rows = cases.truncated_rows(m)
    nnz = int(rp[rows])
    vals, _ = cases.values_for(name, len(ci), n, dtype)
    return api.Tile_create(rows, n, nnz, rp, ci, vals, dtype=dtype, hyb=hyb), rows, n, nnz


def test_options_struct_is_versioned_and_mirrored():
    lib = _lib.load(np.float64)
    o = _lib.PlanOptions()
    raw = (C.c_int * (C.sizeof(o) // 4))()
    lib.tilespmv_plan_options_init(C.cast(raw, C.POINTER(_lib.PlanOptions)))
    init = C.cast(raw, C.POINTER(_lib.PlanOptions)).contents
    assert init.size == C.sizeof(_lib.PlanOptions)                  # same layout on both sides of the ABI
    assert (init.coo_mode, init.dense_mode, init.kernel, init.tilerow_begin, init.tilerow_end, init.autotune) == (0, 0, 0, 0, 0, 0)
    for k in _lib.KNOB_NAMES:
        assert getattr(init, k) == _lib.KNOB_DEFAULT, k
    assert all(v == _lib.KNOB_DEFAULT for v in init.reserved)
    with pytest.raises(TypeError):
        _lib.PlanOptions(no_such_knob=1)


def test_options_mirror_has_the_field_order_of_the_header():
    """the ctypes mirror had mv_native / mv_xcd_chunk / lds_pad / y_store permuted; size and all-defaults checks cannot see that.
    The library reports name:offset of every field as it was compiled; the mirror must agree field by field, and a distinct value written
    through every Python attribute must arrive in the C field of the same name (read back at the C offset)."""
    for dtype in (np.float64, np.float32):
        lib = _lib.load(dtype)
        layout = [kv.split(":") for kv in lib.tilespmv_plan_options_layout().decode().split(",")]
        c_off = {k: int(v) for k, v in layout}
        py_off = {name: getattr(_lib.PlanOptions, name).offset for name, _ in _lib.PlanOptions._fields_}
        assert [k for k, _ in layout] == [name for name, _ in _lib.PlanOptions._fields_]
        assert c_off == py_off
        o = _lib.PlanOptions(**{k: 1000 + i for i, k in enumerate(_lib.KNOB_NAMES)})
        raw = (C.c_int * (C.sizeof(o) // 4)).from_buffer_copy(o)
        for i, k in enumerate(_lib.KNOB_NAMES):
            assert raw[c_off[k] // 4] == 1000 + i, k


@pytest.mark.parametrize("name", ["allfmt", "powerlaw20k", "circuit8k", "one_long_row", "lap64", "kkt12"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_every_entry_mode_packs_and_decodes(name, dtype):
    """The builder re-decodes every packed list in digest builds and fails (-6) on a mismatch: all modes must build."""
    tm, rows, n, nnz = _tm(name, dtype, hyb=(name == "allfmt"))
    seen = set()
    for kw in ({"entry_mode": 0}, {"entry_mode": 1}, {"entry_mode": 2, "wg_strips": 16}, {"entry_mode": 2, "wg_strips": 32},
               {"coo_mode": api.COO_FALLBACK}, {"kernel": api.KERNEL_DIRECT}, {"entry_mode": 2, "strip_cost": 64, "split_above": 200}):
        d, info = api.plan_layout_digest(tm, rows, n, nnz, **kw)
        d2, _ = api.plan_layout_digest(tm, rows, n, nnz, **kw)
        assert d == d2                                   # deterministic (threaded builder, plan-fixed order)
        assert info["nnz"] == nnz and info["rows"] == rows
        seen.add(d)
    assert len(seen) >= 5                                # different knobs really produce different layouts
    api.Tile_destroy(tm)


def test_brick_order_is_detected_on_3d_grids_only():
    """Stride detection + brick order: 3-D stencils (7-point cube, the KKT stand-in) get it by default once they are large,
    2-D stencils and irregular matrices do not; x_window=0 switches it off; the layout stays deterministic."""
    cases_ = {"lap3d96": (G.laplacian7pt(96), True), "kkt48": (G.nlpkkt_like(48, target_nnz=None), True),
              "lap2d1024": (G.laplacian5pt(1024), False), "powerlaw200k": (cases.MEDIUM["powerlaw200k"](), False)}
    for name, ((m, n, rp, ci), want3d) in cases_.items():
        rows = cases.truncated_rows(m); nnz = int(rp[rows])
        tm = api.Tile_create(rows, n, nnz, rp, ci, G.compat_values(len(ci)))
        d_auto, i_auto = api.plan_layout_digest(tm, rows, n, nnz, strip_cost=100)          # (small strips: enough workgroups to count as large)
        d_off, i_off = api.plan_layout_digest(tm, rows, n, nnz, strip_cost=100, x_window=0)
        d_on, i_on = api.plan_layout_digest(tm, rows, n, nnz, strip_cost=100, x_window=2)
        assert i_off["brick_order"] == 0
        assert i_auto["brick_order"] == (1 if want3d else 0), name
        assert i_on["brick_order"] == (1 if name != "powerlaw200k" else 0), name             # forced: wherever strides exist (2-D too)
        assert (d_auto != d_off) == want3d
        assert api.plan_layout_digest(tm, rows, n, nnz, strip_cost=100)[0] == d_auto
        api.Tile_destroy(tm)


def test_dictionary_descriptors_where_patterns_are_few():
    """4-B unit descriptors + a dictionary of column patterns (stencils: a handful of patterns); 12 B when the knob says so,
    when the patterns do not fit the dictionary."""
    m, n, rp, ci = G.laplacian7pt(48)
    rows = cases.truncated_rows(m); nnz = int(rp[rows])
    tm = api.Tile_create(rows, n, nnz, rp, ci, G.compat_values(len(ci)))
    d4, i4 = api.plan_layout_digest(tm, rows, n, nnz)
    d12, i12 = api.plan_layout_digest(tm, rows, n, nnz, desc_dict=0)
    assert (i4["desc_bytes"], i12["desc_bytes"]) == (4, 12) and d4 != d12
    assert i4["nt_stream"] == 0 and api.plan_layout_digest(tm, rows, n, nnz, nt_stream=1)[1]["nt_stream"] == 1    # small launch: default cache policy unless asked
    assert i12["stream_bytes"] - i4["stream_bytes"] >= 8 * (nnz // 16) * 0.9      # 8 bytes per unit less to read
    api.Tile_destroy(tm)
    # an entry-dominated shard with a handful of units: the dictionary would save < 2 % of the streams and is not taken by default (desc_dict=1 asks for it anyway)
    tm, rows, n, nnz = _tm("powerlaw200k")
    _, i_def = api.plan_layout_digest(tm, rows, n, nnz)
    _, i_ask = api.plan_layout_digest(tm, rows, n, nnz, desc_dict=1)
    assert i_def["desc_bytes"] == 12 and i_ask["desc_bytes"] == 4
    api.Tile_destroy(tm)
    # thousands of distinct patterns (random columns inside ELL / CSR-as-units tiles): the dictionary is refused, 12 B stay
    rng = np.random.default_rng(11)
    rows_, cols_ = 16 * 1200, 16 * 1200
    ri = np.repeat(np.arange(rows_), 6)
    cj = (ri // 16) * 16 + rng.integers(0, 16, len(ri))          # six random columns of the diagonal tile per row
    m2, n2, rp2, ci2 = G.from_coo(rows_, cols_, ri, cj)
    tm = api.Tile_create(rows_, cols_, len(ci2), rp2, ci2, G.compat_values(len(ci2)))
    _, info = api.plan_layout_digest(tm, rows_, cols_, len(ci2))
    assert info["desc_bytes"] == 12
    api.Tile_destroy(tm)


def test_rules_that_depend_on_the_size_of_the_launch():
    """Nontemporal value / entry streams above 400 MB per launch, strips of 8 tile-rows (cost target 800) once that leaves >= 4096 workgroups;
    neither on a grid that fits the Infinity Cache."""
    for n, want_nt, want_cost in ((3456, 1, 800), (1024, 0, 400)):
        m, cols, rp, ci = G.laplacian5pt(n)
        rows = cases.truncated_rows(m); nnz = int(rp[rows])
        tm = api.Tile_create(rows, cols, nnz, rp, ci, G.compat_values(len(ci)))
        _, info = api.plan_layout_digest(tm, rows, cols, nnz)
        assert (info["nt_stream"], info["strip_cost"], info["desc_bytes"]) == (want_nt, want_cost, 4), (n, info)
        assert (info["stream_bytes"] > (400 << 20)) == bool(want_nt)
        _, off = api.plan_layout_digest(tm, rows, cols, nnz, nt_stream=0, strip_cost=400)
        assert (off["nt_stream"], off["strip_cost"]) == (0, 400)
        api.Tile_destroy(tm)
        del rp, ci


def test_environment_is_only_a_default(monkeypatch):
    tm, rows, n, nnz = _tm("powerlaw20k")
    by_option, info = api.plan_layout_digest(tm, rows, n, nnz, entry_mode=2, strip_cost=800, entry_ordered=1)
    assert info["entry_mode"] == 2 and info["strip_cost"] == 800 and info["entry_ordered"] == 1
    monkeypatch.setenv("TILESPMV_WAVE_COO", "2"); monkeypatch.setenv("TILESPMV_STRIP_COST", "800"); monkeypatch.setenv("TILESPMV_COO_ORDERED", "1")
    by_env, _ = api.plan_layout_digest(tm, rows, n, nnz)
    assert by_env == by_option
    over, info2 = api.plan_layout_digest(tm, rows, n, nnz, entry_mode=0, strip_cost=400)   # an option beats the environment
    assert info2["entry_mode"] == 0 and info2["strip_cost"] == 400 and over != by_env
    api.Tile_destroy(tm)


def test_two_threads_build_differently_tuned_layouts():
    """No setenv / unsetenv inside the library any more: concurrent builds with different knobs match the serial ones."""
    tm, rows, n, nnz = _tm("powerlaw200k")
    knobs = [dict(entry_mode=i % 3, strip_cost=200 + 250 * i, entry_ordered=i & 1) for i in range(6)]
    serial = [api.plan_layout_digest(tm, rows, n, nnz, **k)[0] for k in knobs]
    out = [None] * len(knobs)

    def work(ids):
        for i in ids:
            out[i] = api.plan_layout_digest(tm, rows, n, nnz, **knobs[i])[0]
    ts = [threading.Thread(target=work, args=(range(j, len(knobs), 3),)) for j in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert out == serial
    api.Tile_destroy(tm)


def test_columns_beyond_2_to_28_keep_their_bits():
    """the round-2 lists kept row bits in the column word's top four bits, so a matrix with more than 2^28
    columns gathered from the wrong x.  Packed records carry (column - chunk base); the builder's decode check covers it."""
    rows, cols = 4096, (1 << 28) + 1000
    rng = np.random.default_rng(3)
    ri = np.repeat(np.arange(rows), 6)
    ci = np.concatenate([rng.integers(0, cols, rows * 3), rng.integers(1 << 28, cols, rows * 3)])
    m, n, rp, cidx = G.from_coo(rows, cols, ri, ci)
    nnz = len(cidx)
    tm = api.Tile_create(rows, cols, nnz, rp, cidx, G.compat_values(nnz))
    d, info = api.plan_layout_digest(tm, rows, cols, nnz, coo_mode=api.COO_FALLBACK)    # tilen > 2^24: first-generation kernel + fallback lists
    assert info["kernel"] == api.KERNEL_DIRECT and info["fallback_nnz"] > 0
    with pytest.raises(RuntimeError):                                                     # the unit stream keeps column blocks in 24 bits: refused, not wrong
        api.plan_layout_digest(tm, rows, cols, nnz, kernel=api.KERNEL_STREAM)
    api.Tile_destroy(tm)
    # and just below the unit stream's limit, with the merged lists of entry modes 1 and 2 (columns up to 2^28 - 1)
    cols = 1 << 28
    ci = np.concatenate([rng.integers(0, cols, rows * 3), rng.integers(cols - 5000, cols, rows * 3)])
    m, n, rp, cidx = G.from_coo(rows, cols, ri, ci)
    nnz = len(cidx)
    tm = api.Tile_create(rows, cols, nnz, rp, cidx, G.compat_values(nnz))
    for em in (1, 2):
        d, info = api.plan_layout_digest(tm, rows, cols, nnz, kernel=api.KERNEL_STREAM, entry_mode=em)
        assert info["entry_mode"] == em
    api.Tile_destroy(tm)
