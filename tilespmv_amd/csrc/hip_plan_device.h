// hip_plan_device.h — the stages of the unit-stream builder (hip_plan_stream.hip) that touch every nonzero, as kernels over a device-resident Tile_matrix
// (hip_tile_create.h DevTile): COUNT and EMIT call the same per-tile functions as the host builder (plan_tile_ops.h), ENCODE's descriptor pass and pattern dictionary
// are a sort + run-length encoding.  The decisions in between (CHOOSE, CUT, ORDER: functions of per-tile-row counts) stay on the host.  SURVEY S8 f1; reference
// src/csr2tile.h:629-1020 is the preprocessing this replaces end to end: with tilespmv_plan_create_from_csr only the CSR arrays cross the bus.
#pragma once
#include <vector>

#include "hip_tile_create.h"
#include "plan_tile_ops.h"

namespace tilespmv {

// per-tile exclusive prefixes of one counting pass over the shard's tiles [t_begin, t_end] (device arrays of t_end - t_begin + 1 ints), + the pooled part per tile-row
struct DevCounts {
    int csr_form = -1;
    int *tu = nullptr, *tc = nullptr, *td = nullptr;   // units / list entries / dense tiles emitted by the tiles before tile t (pool excluded)
    int *tp = nullptr;                                 // pooled plans: nonzeros the tiles before tile t put into the pool
    int *pool_u = nullptr, *pool_c = nullptr;          // pooled plans: units / list entries of every tile-row's pool
    PoolEnt *pool = nullptr;                           // pooled plans: every tile-row's pooled nonzeros, column-major, back to back (filled by COUNT, read again by EMIT)
    void release();
};

struct DevShard { const DevTile *D; int tr0, tr1, t_begin, t_end; bool coo_in_tile, dense_mfma; long long stored0, stored; bool absorb = false, derive = false; };   // absorb: plan_tile_ops.h "absorbed list entries"   // stored: blknnz[t_end] - blknnz[t_begin]

// rc 0 or -3 (HIP error, reported on stderr)
int dev_fetch_ints(const int *d_array, const long long *idx, int n, int *out);   // out[k] = d_array[idx[k]]
int dev_count(const DevShard &S, int csr_form, DevCounts *C, hvec<int> &counts3, long long *pool_units = nullptr, long long *pool_lines = nullptr);   // counts3: (units, list entries, dense tiles) of every tile-row; pool_*: += units made of pooled windows, 128-byte lines of x their gathers touch
// column patterns of the first units on a sample of tiles (what the split form's dictionary would have to hold): ELL slots exactly, of a CSR tile its first unit
int dev_pattern_sample(const DevShard &S, int step, int last_first_tile, int last_rowlen, std::vector<unsigned long long> &patterns);
// EMIT: pu / pc / pd = first unit / list entry / dense tile of every tile-row (ntr + 1), row_k / row_split as in the host builder.  O: device destinations (zeroed by the caller;
// urow is filled with the identity here)
int dev_emit(const DevShard &S, const DevCounts &C, const hvec<long long> &pu, const hvec<long long> &pc, const hvec<long long> &pd, const std::vector<unsigned char> &row_k,
             const std::vector<unsigned char> &row_split, long long NU, const EmitOut &O);
// word 0 of every emitted unit descriptor -> host (brick order scores the column blocks of the strips)
int dev_fetch_word0(const uint4 *d_udesc, long long NU, hvec<unsigned> &w0);
// ENCODE: units of task i move from [map.x, map.x + map.z) to [map.y, ..) of the packed numbering (padding units in between stay zero)
int dev_pack_desc(const uint4 *d_udesc, const uint2 *d_urow, const uint4 *d_ucol, const int4 *d_map, int ntasks, UDesc *d_packed, URow *d_packed_row, uint4 *d_packed_col);   // d_map: device copy of the (old begin, new begin, count) triples
// the distinct (n0, n1) patterns of NUP packed descriptors, ascending, if there are at most `cap` of them (else `over` = true); then the 4-byte form
int dev_shift_histogram(const UDesc *d_packed, long long NUP, unsigned long long hist[8]);   // units per shift code (word 0 >> UNIT_SHIFT_SHIFT)
struct DictRanges { int off[9]; };   // dictionary entries of shift code c (word 0 >> UNIT_SHIFT_SHIFT): [off[c], off[c + 1])
int dev_dict_patterns(const UDesc *d_packed, long long NUP, size_t cap, std::vector<uint4> &dict, DictRanges *ranges, bool *over);   // distinct (shift code, nibbles) patterns in ascending order, as dictionary entries
int dev_compact_desc(const UDesc *d_packed, long long NUP, const uint4 *d_dict, DictRanges ranges, int cb_bits, unsigned *d_compact);
// pooled dictionary plans: the distinct 16-byte patterns (n0, n1, r0, r1) of the packed units, ascending, if at most `cap` (else over); then the 8-byte (word 0, pattern id) form
int dev_pool_dict(const UDesc *d_packed, const URow *d_packed_row, long long NUP, size_t cap, std::vector<uint4> &sorted_patterns, bool *over);
int dev_pool_compact(const UDesc *d_packed, const URow *d_packed_row, long long NUP, const uint4 *d_dict, int ndict, int word_bits, void *d_out);   // word_bits > 0: 4-byte words (hip_plan.h), else 8-byte pairs

// ENTRIES (entry modes 1 / 2): the lists of every group of GS consecutive tasks merged, ordered by column (ties keep task / list order: ONE stable radix sort by
// (group, column) of the entries laid out group by group) and packed (plan_tile_ops.h pack_chunks, the host builder's function) — two passes of one thread per group:
// sizes, then records.  d_cval / d_ccol / d_crow: EMIT's list entries (device); tasks: host copy, coo ranges in EMIT's numbering.
struct DevLists {
    std::vector<int4> wg;           // per group: [record begin, end), first chunk, 0 (host copy: the caller uploads it)
    long long n_rec = 0, n_chunk = 0, scattered = 0;
    ERec *d_rec = nullptr; unsigned *d_base = nullptr;   // scratch until place() has copied them into the plan
    int *d_panel_off = nullptr;      // x_panels > 1: (x_panels + 1) absolute record offsets per group
    std::vector<int> panel_off;      // ... host copy
    void release();
    ~DevLists() { release(); }
};
int dev_entry_lists(const val_t *d_cval, const int *d_ccol, const unsigned char *d_crow, long long NC, const std::vector<STask> &tasks, int GS, int slab_shift, int dest_bits, bool count_scattered,
                    int x_panels, int panel_shift, DevLists *L);

}  // namespace tilespmv
