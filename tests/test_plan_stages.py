"""The unit-stream layout builder runs in stages (tilespmv_amd/csrc/hip_plan_stream.hip: count, choose, cut, emit, order, encode, entries, finish);
tilespmv_plan_layout_stages hashes what each stage produced (host only, no GPU).  These tests pin WHICH stage a knob may change — a knob that
leaks into an earlier stage than it should shows up here, not as a slow or wrong plan on the device."""
import numpy as np
import pytest

import cases
from tilespmv_amd import api, generators as G

ALL = api.STAGE_NAMES


def _tm(gen, dtype=np.float64, hyb=False):
    m, n, rp, ci = gen
    rows = cases.truncated_rows(m); nnz = int(rp[rows])
    return api.Tile_create(rows, n, nnz, rp, ci, G.compat_values(len(ci), dtype), dtype=dtype, hyb=hyb), rows, n, nnz


def _changed(a, b):
    return [k for k in ALL if a[k] != b[k]]


@pytest.fixture(scope="module")
def mats():
    out = {"lap3d": _tm(G.laplacian7pt(48)), "powerlaw": _tm(G.powerlaw(60000, seed=2)), "allfmt": _tm(G.all_formats(12, 7), hyb=True),
           "bandrand": _tm(G.band_plus_random(40000, 4, 3, 5)), "band40": _tm(G.band(30000, 40))}
    yield out
    for tm, *_ in out.values():
        api.Tile_destroy(tm)


def test_stages_are_deterministic_and_all_present(mats):
    for name, (tm, rows, n, nnz) in mats.items():
        a, info = api.plan_layout_stages(tm, rows, n, nnz)
        b, _ = api.plan_layout_stages(tm, rows, n, nnz)
        assert a == b, name
        assert all(a[k] != 0 for k in ALL), (name, a)
        assert len(set(a.values())) == len(ALL)


def test_first_generation_plans_have_no_stages(mats):
    tm, rows, n, nnz = mats["allfmt"]
    a, _ = api.plan_layout_stages(tm, rows, n, nnz, kernel=api.KERNEL_DIRECT)
    assert all(v == 0 for v in a.values())


def test_descriptor_form_touches_the_encoding_and_the_byte_model_only(mats):
    tm, rows, n, nnz = mats["lap3d"]
    d4, i4 = api.plan_layout_stages(tm, rows, n, nnz)
    d12, i12 = api.plan_layout_stages(tm, rows, n, nnz, desc_dict=0)
    assert (i4["desc_bytes"], i12["desc_bytes"]) == (4, 12)
    assert _changed(d4, d12) == ["encode", "finish"]


def test_cache_policy_and_store_knobs_touch_only_the_last_stage(mats):
    tm, rows, n, nnz = mats["lap3d"]
    base, _ = api.plan_layout_stages(tm, rows, n, nnz)
    for kw in ({"nt_stream": 1}, {"y_store": 0}, {"coo_heavy_min": 7}):
        other, _ = api.plan_layout_stages(tm, rows, n, nnz, **kw)
        assert _changed(base, other) == ["finish"], kw
    for kw in ({"xcd_chunk": 8}, {"xcd_remap": 0}, {"lds_pad": 4096}, {"mv_native": 1}):   # launch parameters: no stage at all
        other, _ = api.plan_layout_stages(tm, rows, n, nnz, **kw)
        assert _changed(base, other) == [], kw


def test_ordered_adds_are_a_flag_not_a_layout(mats):
    tm, rows, n, nnz = mats["powerlaw"]
    a, ia = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, entry_ordered=1)
    b, ib = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, entry_ordered=0)
    assert (ia["entry_ordered"], ib["entry_ordered"]) == (1, 0)
    assert _changed(a, b) == ["choose", "finish"]


def test_strips_per_workgroup_only_regroup_the_entry_lists(mats):
    tm, rows, n, nnz = mats["powerlaw"]
    a, _ = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, wg_strips=16, strip_cost=800)
    b, _ = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, wg_strips=32, strip_cost=800)
    assert _changed(a, b) == ["choose", "entries", "finish"]


def test_entry_modes_share_counts_and_differ_in_the_lists(mats):
    tm, rows, n, nnz = mats["bandrand"]
    st = {m: api.plan_layout_stages(tm, rows, n, nnz, entry_mode=m, strip_cost=800, split_above=100000)[0] for m in (0, 1, 2)}
    assert st[0]["count"] == st[1]["count"] == st[2]["count"]
    assert st[1]["cut"] == st[2]["cut"] and st[1]["emit"] == st[2]["emit"] and st[1]["encode"] == st[2]["encode"]   # same strips, same units: only the merged lists differ
    assert st[1]["entries"] != st[2]["entries"] != st[0]["entries"]


def test_brick_order_permutes_tasks_but_emits_the_same_units(mats):
    tm, rows, n, nnz = mats["lap3d"]
    kw = dict(strip_cost=100)
    lin, il = api.plan_layout_stages(tm, rows, n, nnz, x_window=0, **kw)
    brk, ib = api.plan_layout_stages(tm, rows, n, nnz, x_window=2, **kw)
    assert (il["brick_order"], ib["brick_order"]) == (0, 1)
    assert lin["count"] == brk["count"]
    assert lin["order"] != brk["order"] and lin["encode"] != brk["encode"]
    one, i1 = api.plan_layout_stages(tm, rows, n, nnz, x_window=1, **kw)      # (1 was the LDS-window form, retired in round 6: it now means what 2 means)
    assert one == brk and i1["brick_order"] == 1


def test_dense_tiles_as_units_or_for_the_matrix_cores(mats):
    tm, rows, n, nnz = mats["band40"]
    u, iu = api.plan_layout_stages(tm, rows, n, nnz, dense_mode=api.DENSE_VALU)
    m, im = api.plan_layout_stages(tm, rows, n, nnz, dense_mode=api.DENSE_MFMA)
    assert u["count"] != m["count"] and u["emit"] != m["emit"]
    assert im["stream_bytes"] < iu["stream_bytes"]      # 256 values + 4 B per dense tile against 16 units of 16 values + descriptor


def test_byte_model_follows_the_streams(mats):
    """stream_bytes (the plan's own model of one SpMV) recomputed from the plan facts for the two simplest layouts."""
    tm, rows, n, nnz = mats["lap3d"]          # units only (7-point grid aligned to 16: no entries), 4-B descriptors, no padding units beyond the value groups
    _, i = api.plan_layout_stages(tm, rows, n, nnz, desc_dict=0, strip_even=0)
    units = nnz // 16 if nnz % 16 == 0 else None
    assert i["stream_bytes"] >= nnz * 8 + 12 * (nnz // 16) + i["num_tasks"] * 32 + 8 * (rows + n)
    assert i["stream_bytes"] <= 1.1 * (nnz * 8 + 12 * (nnz // 16) + i["num_tasks"] * 32 + 8 * (rows + n))
    tm, rows, n, nnz = mats["powerlaw"]       # entry-dominated, per-strip lists: value + column + row byte per entry
    _, i0 = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=0)
    _, i2 = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2)
    assert i0["stream_bytes"] >= 13 * int(0.8 * nnz) and i2["stream_bytes"] < i0["stream_bytes"]     # packed 12-B records against 13-B triples


def test_column_panels_are_offsets_into_the_same_lists(mats):
    """Panels are recorded as offsets into the one column-ordered list of a group: the records do not change, and how many panels a pass takes — or whether the XCDs take them as
    column slices — is a launch fact.  (Putting a group's entries around its own rows first and letting the panels divide only the rest was tried: slower in all three forms.)"""
    tm, rows, n, nnz = mats["bandrand"]
    one, i1 = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=0)
    many, ik = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_panel_merge=1)
    assert (i1["x_panels"], ik["x_panels"]) == (1, -(-n // 4096))
    assert _changed(one, many) == ["choose", "entries"]
    m2, i2 = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_panel_merge=2)
    assert _changed(many, m2) == [] and i2["x_panels"] == -(-ik["x_panels"] // 2)
    assert ik["stream_bytes"] > i2["stream_bytes"] > i1["stream_bytes"]          # the passes beyond the first read and write their rows of y
    off, io = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_panel_merge=0)
    assert _changed(many, off) == [] and io["x_panels"] == 1 and io["stream_bytes"] == i1["stream_bytes"]
    sl, isl = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_slice_passes=2)
    assert _changed(many, sl) == [] and isl["x_slice_passes"] == 2 and isl["x_panels"] == 2 and isl["x_panel_merge"] == 0 and isl["entry_ordered"] == 0
    so, iso = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_slice_passes=2, entry_ordered=1)
    assert iso["x_slice_passes"] == 0 and iso["entry_ordered"] == 1                # reproducible sums asked: the sliced form (atomic adds in any order) stays out
    for kw in (dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2, wg_strips=32)):
        assert api.plan_layout_stages(tm, rows, n, nnz, x_panel_kb=32, x_panel_merge=1, **kw)[1]["x_panels"] == 1, kw


def test_pooled_units_are_chosen_by_the_byte_model_and_are_a_layout_of_their_own():
    """Round 5: CSR-format tiles as pooled units (csr_split = 2).  Unset, the form is the one that puts fewer bytes into the streams: pooled on a block-structured (FEM-like) shard whose
    nonzeros sit in ragged CSR tiles, the ELL-style split on stencil-like shards whose units share a few column patterns (4-byte dictionary descriptors).  The form is decided in the
    count stage, so every later stage follows; pooled plans have 20-byte descriptors — or a dictionary of 16-byte patterns where the shard's units use at most 1,024 of them (natural-order meshes: a few dozen), with one 4-byte
    word per unit (round 6; 8-byte pairs where base, id and tile-row do not fit a word, or with desc_dict = 2) —, 4-row strips and no column panels."""
    tm, rows, n, nnz = _tm(G.fem_hex(12, 12, 12, 3))
    auto, ia = api.plan_layout_stages(tm, rows, n, nnz)
    split, is_ = api.plan_layout_stages(tm, rows, n, nnz, csr_split=1)
    pooled, ip = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2)
    whole, iw = api.plan_layout_stages(tm, rows, n, nnz, csr_split=0)
    assert (ia["csr_form"], is_["csr_form"], ip["csr_form"], iw["csr_form"]) == (2, 1, 2, 0)
    assert auto == pooled and {k: v for k, v in ia.items() if not k.endswith("_us")} == {k: v for k, v in ip.items() if not k.endswith("_us")}
    assert ip["stream_bytes"] < 0.93 * is_["stream_bytes"] and ip["desc_bytes"] == 4
    pairs, ipair = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2, desc_dict=2)
    assert ipair["desc_bytes"] == 8 and ipair["stream_bytes"] > ip["stream_bytes"] and set(_changed(pairs, pooled)) <= {"encode", "entries", "finish"} and "encode" in _changed(pairs, pooled)   # (the same units, another encoding)
    nodict, ind = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2, desc_dict=0)
    assert ind["desc_bytes"] == 20 and set(_changed(nodict, pooled)) <= {"encode", "entries", "finish"} and "encode" in _changed(nodict, pooled) and ind["stream_bytes"] > ip["stream_bytes"]   # the dictionary is an encoding of the same units (the entries stage hashes its uploads on top of the running digest)
    assert _changed(split, pooled) == ALL
    b_alg = api.algorithmic_bytes(nnz, rows, n, 8)
    assert ip["stream_bytes"] < 0.85 * b_alg            # values + 1.25 bytes per slot, fill > 0.9 even on this small mesh (boundary rows are a third of it)
    for kw in (dict(entry_mode=0), dict(entry_mode=1), dict(entry_mode=2), dict(entry_mode=2, x_panel_kb=4, x_panel_merge=1), dict(desc_dict=1), dict(x_window=1), dict(entry_mode=2, wg_strips=32)):
        _, i = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2, **kw)
        assert i["csr_form"] == 2 and i["desc_bytes"] == 4 and i["x_panels"] == 1 and i["wg_strips"] == 16, kw
    api.Tile_destroy(tm)
    # (round 6: stencils on grids whose lines are not a multiple of 16 long leave a tenth and more of their nonzeros in small COO tiles — the off-diagonals cross the tile boundaries at a
    #  different row in every tile-row —; pooled windows start at any column and take them all: 5-point 200^2 is pooled now, the aligned 256^2 and 48^3 grids stay classic)
    for gen, want in ((G.laplacian7pt(48), 1), (G.laplacian5pt(256), 1), (G.laplacian5pt(200), 2), (G.fem_hex(9, 9, 9, 6), 2), (G.fem_hex(14, 11, 9, 3, shuffle=16), 2)):
        tm, rows, n, nnz = _tm(gen)
        assert api.plan_layout_stages(tm, rows, n, nnz)[1]["csr_form"] == want
        api.Tile_destroy(tm)
    for dtype in (np.float32,):
        tm, rows, n, nnz = _tm(G.fem_hex(12, 12, 12, 3), dtype)
        _, i = api.plan_layout_stages(tm, rows, n, nnz)
        assert i["csr_form"] == 2 and i["stream_bytes"] < 0.80 * api.algorithmic_bytes(nnz, rows, n, 4)
        api.Tile_destroy(tm)


def test_wide_windows_are_taken_where_their_gathers_stay_on_few_lines():
    """Round 5 (second half): wide pooled units (csr_form 3: windows of 256 columns, a byte of column offset per slot).  Counted beside the 16-column form and taken where they move at
    least 4 % of the nonzeros off the entry lists while a unit's gathers touch at most POOL_WIDE_MAX_LINES lines of x on average: window-shuffled FEM meshes yes, natural-order meshes no
    (their 16-column windows are full and have the pattern dictionary), a mesh shuffled over thousands of nodes no (every slot on a line of its own)."""
    for gen, want in ((G.fem_hex(16, 16, 16, 3, shuffle=64), 3), (G.fem_hex(12, 12, 12, 6, shuffle=64), 3), (G.fem_hex(12, 12, 12, 3), 2), (G.tet_mesh(30, shuffle=512), 1), (G.laplacian7pt(32), 1), (G.laplacian7pt(40), 2)):
        tm, rows, n, nnz = _tm(gen)
        st, i = api.plan_layout_stages(tm, rows, n, nnz)
        assert i["csr_form"] == want, (rows, nnz, i["csr_form"], want)
        if want == 3:
            assert i["desc_bytes"] == 28
            forced, i2 = api.plan_layout_stages(tm, rows, n, nnz, csr_split=3)
            assert forced == st
            p16, i16 = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2)
            assert i["stream_bytes"] <= 1.03 * i16["stream_bytes"]
        api.Tile_destroy(tm)


def test_pooled_plans_keep_every_nonzero_once(mats):
    """The pooled form re-distributes nonzeros between units and list entries; whatever the knobs, units x 16 + list entries must cover the stored nonzeros (the GPU suite checks the values)."""
    tm, rows, n, nnz = mats["allfmt"]
    for kw in (dict(), dict(coo_mode=api.COO_FALLBACK), dict(dense_mode=api.DENSE_MFMA), dict(entry_mode=0, strip_cost=64, split_above=200)):
        a, ia = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2, **kw)
        b, ib = api.plan_layout_stages(tm, rows, n, nnz, csr_split=2, **kw)
        assert a == b and ia["csr_form"] == 2


def test_deterministic_switch_turns_every_timed_choice_off(mats):
    """tilespmv_plan_options.deterministic: no stopwatch (placement retry, column panels / slices, pacing calibration) and ordered sums; a layout fact, visible without a GPU."""
    tm, rows, n, nnz = mats["bandrand"]
    base, ib = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32)
    det, idt = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, deterministic=1)
    assert idt["entry_ordered"] == 1 and idt["x_panel_merge"] == 0 and idt["x_slice_passes"] == 0
    sl, isl = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_slice_passes=2, entry_ordered=0, deterministic=1)
    assert isl["x_slice_passes"] == 0 and isl["entry_ordered"] == 1      # the switch wins over knobs that would make the sums meet in any order
    pm, ipm = api.plan_layout_stages(tm, rows, n, nnz, entry_mode=2, x_panel_kb=32, x_panel_merge=2, deterministic=1)
    assert ipm["x_panel_merge"] == 2                                      # a FIXED panelled form times nothing and adds in a fixed order: allowed


def test_absorbed_entries_are_decided_in_the_count_stage(mats):
    """Round 6: the corner entries of a stencil's COO tiles move into the padding of the neighbouring ELL units (tilespmv_plan_options.absorb, csrc/plan_tile_ops.h).  The rule is a
    per-tile function used by COUNT (fewer list entries) and EMIT (shifted units) alike, so every stage from the first one on sees it; on the aligned 7-point grid no list entry is left,
    the streams shrink, and the descriptor form (4-byte words + dictionary, now keyed by shift and nibbles) stays what it was.  Off: the plan of round 5, stage for stage."""
    tm, rows, n, nnz = mats["lap3d"]
    on, ion = api.plan_layout_stages(tm, rows, n, nnz)
    off, ioff = api.plan_layout_stages(tm, rows, n, nnz, absorb=0)
    dflt, idf = api.plan_layout_stages(tm, rows, n, nnz, absorb=1)
    assert on == dflt and ion["list_entries"] == idf["list_entries"]
    gather, ig = api.plan_layout_stages(tm, rows, n, nnz, absorb=2)   # absorbed entries, but no derived units: the same counts and cuts, other descriptors
    assert ion["derived_units"] > 0 and ig["derived_units"] == 0 and ioff["derived_units"] == 0 and ig["list_entries"] == ion["list_entries"]
    assert _changed(on, gather) == ["emit", "encode", "finish"] or _changed(on, gather) == ["emit", "encode"], _changed(on, gather)
    assert "count" in _changed(on, off) and "emit" in _changed(on, off) and "encode" in _changed(on, off)
    assert ioff["list_entries"] > 0 and ion["list_entries"] == 0 and ion["stream_bytes"] < ioff["stream_bytes"]
    assert (ion["desc_bytes"], ioff["desc_bytes"]) == (4, 4)
    # nothing to absorb where COO tiles go to the CSR fallback, or in a matrix without ELL tiles next to COO tiles
    a, ia = api.plan_layout_stages(tm, rows, n, nnz, coo_mode=api.COO_FALLBACK)
    b, ib = api.plan_layout_stages(tm, rows, n, nnz, coo_mode=api.COO_FALLBACK, absorb=0)
    assert a == b and ia["list_entries"] == ib["list_entries"]
    tm2, rows2, n2, nnz2 = mats["band40"]
    a, _ = api.plan_layout_stages(tm2, rows2, n2, nnz2)
    b, _ = api.plan_layout_stages(tm2, rows2, n2, nnz2, absorb=0)
    assert a == b
