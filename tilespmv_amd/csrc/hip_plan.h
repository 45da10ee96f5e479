// hip_plan.h — device-resident layout of one Tile_matrix shard ("plan") for gfx950.
//
// Data layout in HBM (all streams are in tile order = tile-row major, so the bytes one
// workgroup consumes are one contiguous run of each stream):
//
//   desc[]  uint2 per emitted tile   .x = column block, .y = packed meta (DESC_* below)
//   val[]   val_t                    every tile's payload values back to back, in the tile's
//                                    own format layout (row stride always 16, zero padded)
//   idx[]   bytes                    every tile's index bytes back to back (tile-local packing)
//   task[]  Task                     one per 16-lane lane group: a strip of whole tile-rows,
//                                    or one piece of a split (very long) tile-row
//
// Formats keep the reference's semantics; only WHERE the bytes sit
// changes (the reference keeps one array per format, src/format.h:26-50).
#pragma once
#include <cstdint>

#include "host_util.h"

namespace tilespmv {

// desc.y bit fields
constexpr unsigned DESC_FMT_MASK = 7u;    // TILESPMV_FMT_* ; 7 = no-op (empty tile-row)
constexpr unsigned DESC_FMT_NOP = 7u;
constexpr unsigned DESC_EOR = 8u;         // last tile of its tile-row
constexpr int DESC_P1_SHIFT = 4;          // 8 bits: CSR nnz | COO count | ELL/HYB width | #dense rows/cols
constexpr int DESC_P2_SHIFT = 12;         // 8 bits: HYB remainder count

// ---- packed entry record (merged, column-ordered entry lists of the wavefront / workgroup entry modes and of the fallback):
// value + one index word = (column - chunk base) << dest_bits | destination row of the group.  One lane load per entry
// (global_load_dwordx3 in fp64, dwordx2 in fp32) instead of three stream loads (value, column, row byte): 12 B instead of
// 13 B per entry in fp64, 8 instead of 9 in fp32, and — what the ablations of round 3 showed to matter as much as the
// bytes — one vector-memory instruction per 64 entries instead of three (profiles/r03_entry_ablations.txt).  The column
// base is one 32-bit word per chunk of 64 consecutive records of a list (a wavefront's share of one sub-trip), fetched
// with a scalar load.  A chunk whose columns span more than 2^(32 - dest_bits) is cut short and padded with null records
// (value 0, offset 0, destination 0).  Columns are full 32-bit numbers again (the round-2 lists kept the strip in the
// column word's top four bits, which limited them to matrices of < 2^28 columns).
#if defined(TILESPMV_F32)
struct ERec { unsigned v, w; };            // 8 bytes
#else
struct ERec { unsigned lo, hi, w; };       // 12 bytes, 4-byte aligned
#endif
constexpr int ECHUNK = 64;                 // records per column-base chunk
#ifndef TILESPMV_SMALL_GRID_WORKGROUPS
#define TILESPMV_SMALL_GRID_WORKGROUPS 128
#endif
constexpr int SMALL_GRID_WORKGROUPS = TILESPMV_SMALL_GRID_WORKGROUPS;   // grids with fewer 256-thread workgroups than this (half the CUs) run 128-thread workgroups of 8 strips (hip_kernels.hip launch_tiles_stream):
                                                                    // 33-72 workgroups 6-10 % faster, 157-247 (5-pt 400^2, the scircuit stand-in) within 1.5 % either way — profiles/r05_small_grid_forms.txt
constexpr int FB_DEST_BITS = 11;           // fallback row blocks: <= 2048 rows

struct Task {
    int tile_begin, tile_end;  // range in desc[]
    long long val_off;         // element offset of tile_begin's payload in val[]
    long long idx_off;         // byte offset in idx[]
    int row;                   // tile-row of tile_begin (global numbering)
    int partial;               // -1: whole tile-rows, results go to y; >=0: slot in partial[]
};

struct FixRow { int row, first, count, pad; };  // split tile-row: partial[first .. first+count) -> y

struct DevPlan {
    const uint2 *desc;
    const val_t *val;
    const unsigned char *idx;
    const Task *task;
    int ntasks;
    int rowA, colA;
    val_t *partial;
    const FixRow *fix;      // every split tile-row (first-generation kernel, multi-vector path)
    int nfix;
    const FixRow *fix_late; // split tile-rows with pieces outside the unit kernel: summed by k_fixup_split after all passes
    int nfix_late;
    // very-sparse fallback (the extracted matrix of the shard's rows, deferredcoo_*): row blocks of <= FB_ROWS rows and
    // <= FB_CAP nonzeros, one workgroup each, the block's nonzeros ordered by column and stored as packed entry records
    // (ERec below; destination = row-in-block, FB_DEST_BITS bits); a block's list starts on a chunk boundary
    const int4 *f_blk;      // per block: first local row, #rows (-1: one piece of a single row longer than FB_CAP -> atomic add), [begin, end) in f_rec
    const ERec *f_rec;
    const unsigned *f_base; // column base per 64-record chunk (chunk of record i = i >> 6)
    int f_nblk;
    int f_ordered;          // wavefronts add in turn (bit-reproducible sums)
    int f_row0;             // first global row of the shard
    int f_rows;
};

// ---- second-generation layout: the "unit stream"  -------------------------------
// A unit is one 16-value column of a tile (one ELL slot, one dense / dense-col column): lane r
// of a strip owns row r.  Every unit is self-describing (16-B descriptor: column block, the 16
// column nibbles, flags) and units are addressed by index alone, so the payload loads of the
// next units never wait on decoding the previous tile.  COO tiles (and HYB remainders) of a
// strip are flattened into an entry list with global column ids; CSR / dense-row tiles (and
// dense tiles when they run on the matrix cores) stay whole tiles in a "heavy" list that
// keeps the first-generation per-tile layout (DevPlan streams) and is executed by the
// first-generation kernel in accumulate mode after the unit kernel has written y.
constexpr int FB_ROWS = 2048;             // fallback row block: rows (= 16 x 128: the entry lists' 4 + 7 destination bits) ...
constexpr int FB_CAP = 6144;              // ... and nonzeros (4 trips of 6 x 256)
constexpr int STRIP_MAX_ROWS = 8;         // tile-rows per strip (3 bits of row-in-strip)
constexpr unsigned UNIT_EOR = 1u;         // unit flag bit 0: last unit of its tile-row -> write y
constexpr int UNIT_ROW_SHIFT = 1;         // unit flag bits 1-3: tile-row inside the strip
constexpr unsigned UNIT_ROWUNIT = 16u;    // unit flag bit 4: "row unit" = 16 values of ONE tile row (dense-row tiles);
                                          //   lane = column, word 1/3 hold the target row, result needs a 16-lane reduction
constexpr int UNIT_SHIFT_SHIFT = 29;      // unit flag bits 5-7 (word 0 bits 29-31): signed window shift of a unit that took list entries (plan_tile_ops.h "absorbed list entries"):
constexpr int ABSORB_SHIFT_MIN = -3, ABSORB_SHIFT_MAX = 3;   //   x index = column block * 16 + shift + nibble.  Dictionary plans keep the shift with the pattern (DevStream::udict)
constexpr unsigned UNIT_DERIVED_CODE = 4u;                   //   shift code -4: a DERIVED unit — lanes 0-14 take the previous unit's x one lane up, lane 15 loads (plan_tile_ops.h)
constexpr int UNIT_GROUP = 16 / (int)sizeof(val_t);  // units whose values share one 16-byte lane load (2 in fp64, 4 in fp32)
constexpr long long NT_STREAM_MIN_BYTES = 400ll << 20;   // launches that move more than this (about 1.6 x the 256 MB Infinity Cache) read their once-read streams nontemporally
constexpr int DICT_MAX_BITS = 10;         // dictionary plans: at most 1024 column patterns (8 KB: stays in the vector L1)
constexpr int UNIT_FLAG_SHIFT = 24;       // flags live above the 24-bit column block in words 0 and 2

struct UDesc { unsigned w0, n0, n1; };    // 12 bytes in HBM (w0 = column block | flags << 24: end of row, row in strip, row unit, window shift; n0 / n1 = column nibbles of rows 0-7 / 8-15); lanes expand it to (w0, n0, w0, n1) in LDS so that a lane reads one 8-B half

// ---- pooled units (round 5; the execution form of CSR-format tiles on block-structured / FEM-like shards, where > 90 % of the nonzeros sit in ragged CSR tiles:
// reference pack src/csr2tile.h:429-451, GPU routine src/tilespmv_cuda.h:531-561).  The nonzeros of a tile-row's CSR tiles, COO tiles and HYB remainders are POOLED in
// column-major order (column, then row) and cut into units of up to 16 of them whose columns fall into one window [base, base + 16) of x: slot s of a unit = value,
// column offset nibble, row nibble.  A unit costs 12 (UDesc: w0 = base | tile-row-in-strip << 28, the 16 column nibbles) + 8 (URow: the 16 row nibbles) + 16 values;
// only the last unit of a run of adjacent columns is partly empty (fill 0.99 on a 27-point hex mesh with 3 dof per node), so the streams come to about s_v + 1.3 bytes
// per nonzero — against the s_v + 1 per STORED slot + 12 per unit of the ELL-style split (w padded units per tile, the rest as 13-byte list entries), which on ragged
// tiles stores 1.3-1.6 slots per nonzero.  Lanes no longer own rows: every product goes to the strip's slab of s_y with ds_add_f64 (destination = tile-row-in-strip,
// row nibble) and y is stored from the slab.  In a pooled plan EVERY unit has this form (ELL slots, dense and dense-col columns: base = 16 x column block, identity row
// nibbles; dense-row units: identity column nibbles, one row nibble), so the kernel has one code path.  Windows that hold fewer nonzeros than a unit is worth
// (POOL_MIN_FILL) stay on the strip's entry list.
// Pooled DICTIONARY plans (round 5, second half): where the units of a shard use at most 2^DICT_MAX_BITS distinct 16-byte patterns (16 column nibbles + 16 row nibbles) — natural-order
// meshes use a few dozen: 54 on the 27-point hex mesh x 3 unknowns, 31 on the tetrahedral mesh — a unit's descriptor in HBM is 8 bytes (w0, pattern id) and the patterns sit in
// DevStream::pdict, which stays in the vector L1 / L2: 136 instead of 148 bytes per unit (streams -8 %, time -2 ... -5 %: profiles/r05_pool_dictionary_ab.txt).  desc_dict = 0 keeps the
// 20-byte form; window-shuffled meshes (10^5 patterns) keep it by themselves.
// Round 6: where window base, pattern id and tile-row fit one word — base < 2^b with b = 30 - (bits of the largest id), i.e. up to 2^24 columns with 64 patterns, 2^20 with 1024 — the
// descriptor is that 4-byte word: base | id << b | tile-row in strip << 30 (DevStream::cb_bits = b; 0 = the 8-byte pairs): 132 bytes per unit.  desc_dict = 2 keeps the 8-byte pairs.
constexpr int POOL_KR_SHIFT = 28;         // w0 of a pooled unit: first column of the window (28 bits: the unit path already limits shards to 2^24 column blocks) | tile-row in strip << 28
constexpr unsigned POOL_BASE_MASK = (1u << POOL_KR_SHIFT) - 1u;
constexpr int POOL_WORD_KR_SHIFT = 30;    // 4-byte pooled dictionary descriptors: tile-row in strip in the top two bits
__host__ __device__ inline int pool_word_base_bits(int npatterns)   // bits left for the window base once the ids of `npatterns` patterns and the tile-row are in the word
{
    int ib = 1;
    while ((1 << ib) < npatterns) ib++;
    return POOL_WORD_KR_SHIFT - ib;
}
constexpr int POOL_MIN_FILL = sizeof(val_t) == 8 ? 12 : 10;   // 16 s_v + 20 bytes per unit against s_v + 5 (4 in the packed lists) per list entry
#ifndef TILESPMV_POOL_STRIP_ROWS
#define TILESPMV_POOL_STRIP_ROWS 4
#endif
constexpr int POOL_STRIP_ROWS = TILESPMV_POOL_STRIP_ROWS;        // tile-rows per strip of a pooled plan (their tile-rows are heavy; the slab of s_y is half the size: 14.5 KB of LDS per workgroup)
struct URow { unsigned r0, r1; };         // row nibbles of slots 0-7 / 8-15 (slot 0 in the top nibble)
// Wide pooled units (round 5, second half; csr_form 3): the same pooling with windows of POOL_WIDE_WINDOW columns — a slot's column offset is a byte instead of a nibble (16 bytes per unit in
// DevStream::ucol, slot s in byte s; the descriptor's two nibble words then hold the ROW nibbles and there is no urow stream): 28 + 16 s_v bytes per unit.  For shards whose nonzeros are
// spread inside a few hundred columns around their neighbours — window-shuffled meshes, circuit-like and web-graph-like local parts —, where 16-column windows leave 20-90 % of the nonzeros
// on the 12-13-byte entry lists and 256-column windows take most of them at 9.75 bytes each.
constexpr unsigned POOL_WIDE_WINDOW = 256;
constexpr double POOL_WIDE_MAX_LINES = 3.5;   // wide windows are taken only where a unit's 16 gathers touch at most this many 128-byte lines of x on average (hip_plan_stream.hip count())

struct STask {                            // 32 bytes
    int unit_begin, unit_end;
    int coo_begin, coo_end;
    int row;                              // first tile-row of the strip (global numbering)
    int partial;                          // -1 or slot in partial[]
    unsigned nounit_mask;                 // bit k: tile-row k of the strip has no unit (flushed at the end);
                                          //   pieces (partial >= 0): index into DevStream::ifix, or 0xFFFFFFFF
    int nrows;
};

struct DevStream {
    const UDesc *udesc;                   // per unit, 12 B: column block | flags << 24, column nibbles of rows 0-7, of rows 8-15 (dictionary plans: 4-B words, see cb_bits)
    const URow *urow;                     // pooled plans: the row nibbles of every unit (nullptr otherwise, and in pooled dictionary plans)
    const uint4 *ucol;                    // wide pooled plans (csr_form 3): the column-offset bytes of every unit (nullptr otherwise)
    const uint4 *pdict;                   // pooled dictionary plans: udesc holds 8-byte (word 0, pattern id) pairs — or, cb_bits > 0, 4-byte words —, pdict[id] = column nibbles 0-7, 8-15, row nibbles 0-7, 8-15 (nullptr: none)
    int pooled;                           // 1: every unit is a pooled unit (above)
    const val_t *uval;                    // 16 values per unit, stored in groups of UNIT_GROUP units of one task, interleaved per row
    const val_t *cval;                    // COO entry list: value, global column, (row-in-strip << 4) | row
    const int *ccol;
    const unsigned char *crow;
    const STask *task;
    int ntasks;
    int coo_heavy_min;                    // entry mode 0: strips with more COO entries than this run their entry list before the unit pipeline
    int coo_ordered;                      // workgroup entry mode: wavefronts add in turn (bit-reproducible sums)
    int y_streaming;                      // y stores carry the nontemporal hint (plans whose y is a real share of the traffic) or are plain
    // entry modes 1 / 2 (k_units<.., 1 | 2>): the entries of the 4 strips of one wavefront / the 16 or 32 strips of one
    // workgroup, merged and ordered by column, so that the lanes of one gather share x lines; packed records (ERec),
    // destination = strip-in-group << 7 | row-in-strip << 4 | row = the index into the group's slabs of s_y
    const int4 *wg_coo;                   // per group (wavefront or workgroup): [begin, end) in grec, first chunk in gbase, 0
    const ERec *grec;
    const unsigned *gbase;                // column base per chunk of ECHUNK records, chunks counted from the list's begin
    int dest_bits;                        // 9 (wavefront lists), 11 (16 strips per workgroup) or 12 (32 strips)
    const uint4 *udict;                   // dictionary plans: the column patterns (nibbles of rows 0-7, of rows 8-15, window shift << UNIT_SHIFT_SHIFT, 0), ordered by (shift, nibbles); udesc / udesc_cb then hold 4-B words
    int cb_bits;                          // ... column block (cb_bits) | pattern id | flags << 27;  0 = 12-B descriptors.  Pooled dictionary plans: bits of the window base in the 4-byte word, 0 = 8-byte pairs
    int nt_stream;                        // 1: value / entry-record loads are nontemporal (the plan's streams do not fit the Infinity Cache)
    // column panels (round 4): a group's merged list is in column order, so the entries of column panel p (2^k columns, a few MB of x) are the run
    // [panel_off[group * (x_panels + 1) + p], panel_off[.. + p + 1]) of it.  A launch either walks whole lists in k_units (panel_merge = 0) or gives k_units the first
    // panel_merge panels and each further run of panel_merge panels a launch of k_entries_acc (y +=): all gathers of one pass then fall into one slice of x.
    const int *panel_off;                 // nullptr: no panels recorded
    int x_panels;                         // finest panels recorded (1: none)
    int panel_merge;                      // panels per pass of the panelled form; 0 = whole lists in k_units (chosen by timing at plan creation, hip_plan.hip)
    int n_groups;                         // groups (workgroups of the entry phase)
    // column slices pinned to XCDs (round 4, hip_kernels.hip k_entries_xcd): the other use of panel_off.  k_units leaves the lists alone; a launch of 8 x n_groups workgroups
    // follows per pass, workgroup b (dispatched to XCD b & 7) takes group b >> 3's entries of column slice (pass * 8 + (b & 7)) of 8 * slice_passes slices, so one XCD only
    // ever gathers from its own slice of x — which stays in its L2 — and adds the rows it touched to y atomically (sums not bit-reproducible).  0 = off
    int slice_passes;
    int slice_ct;                         // ... records per lane and trip of k_entries_xcd (4, 6 or 8: set with slice_passes from the average run length)
    const UDesc *udesc_cb;                // descriptors of the multi-vector kernel (== udesc)
    // split tile-rows whose pieces all live in the unit kernel are summed in that kernel by the piece that
    // finishes last (fixed slot order): ifix[i] describes row i, ifix_count[i] counts finished pieces
    const FixRow *ifix;
    unsigned *ifix_count;
#ifdef TILESPMV_STAMPS
    unsigned long long *stamps;           // diagnostic build only (scripts/stamps_probe.py): 8 clock stamps per wavefront
#endif
};

// Dense tiles on the matrix cores (generation 2): one wavefront per tile-row that owns dense tiles.
struct DenseRow { int row, tile_begin, tile_end, partial; };  // partial: -1 -> y += result, else slot in partial[]
// Where tile[r][c] of a dense tile sits among its 256 values (matrix-core pass).  Lane (q = c >> 2, r) needs columns 4q .. 4q+3 of row r.  fp32: its 16 bytes
// contiguous — one load per lane, the wavefront's load covers whole lines.  fp64: two halves of 16 bytes, 1 KB apart, so that EACH of the lane's two loads covers whole
// lines too (round 3: with the 32 bytes contiguous every load touched half of every line, which nontemporal loads then fetched twice — band hbw 40 0.265 -> 0.301 ms).
__host__ __device__ constexpr int dense_slot(int r, int c)
{
    return sizeof(val_t) == 8 ? ((c & 3) >> 1) * 128 + ((c >> 2) * 16 + r) * 2 + (c & 1) : ((c >> 2) * 16 + r) * 4 + (c & 3);
}

struct DevDense {
    const int *cb;          // column block per dense tile
    const val_t *val;       // 256 values per tile, zero padded, MFMA operand order: tile[r][c] at dense_slot(r, c)
    const DenseRow *rows;
    int nrows;
};

inline void tile_stream_sizes(int fmt, int p1, int p2, int *nv, int *ni)
{
    switch (fmt) {
    case TILESPMV_FMT_CSR: *nv = p1; *ni = 16 + (p1 + 1) / 2; break;
    case TILESPMV_FMT_COO: *nv = p1; *ni = p1; break;
    case TILESPMV_FMT_ELL: *nv = 16 * p1; *ni = 8 * p1; break;
    case TILESPMV_FMT_HYB: *nv = 16 * p1 + p2; *ni = 8 * p1 + p2; break;
    case TILESPMV_FMT_DNS: *nv = 256; *ni = 0; break;
    case TILESPMV_FMT_DNSROW: *nv = 16 * p1; *ni = p1; break;
    case TILESPMV_FMT_DNSCOL: *nv = 16 * p1; *ni = p1; break;
    default: *nv = 0; *ni = 0; break;
    }
}

}  // namespace tilespmv
