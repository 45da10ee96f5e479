// plan_tile_ops.h — what ONE source tile (and, for pooled plans, one tile-row's pool) becomes in the unit stream: counts and emitted records.
// Written once for the host builder (hip_plan_stream.hip, loops over tile-rows on host threads) and the device builder (hip_plan_device.hip, one thread per tile /
// per tile-row): both call THESE functions, so the streams come out byte for byte the same.  No allocation, no std:: containers; scratch comes from the caller.
#pragma once
#include <hip/hip_runtime.h>

#include "hip_plan.h"

namespace tilespmv {

TILESPMV_HD inline int nib_at(const unsigned char *s, long long p) { return (p & 1) ? (s[p >> 1] & 15) : (s[p >> 1] >> 4); }

// A CSR tile is executed as w ELL-style units (the first w entries of every row) plus the rest of its entries on the strip's COO list; w minimises the bytes moved
// (HYB's idea, src/csr2tile.h:279-306, with this kernel's byte costs).  Returns w and the number of remainder entries.
TILESPMV_HD inline int csr_split_width(const unsigned char *ptr, int rowlen, int nnz, int *remainder)
{
    const long long unit_b = 16 + 16 * (long long)sizeof(val_t), entry_b = (long long)sizeof(val_t) + 5;
    int len[16], wmax = 0;
    for (int r = 0; r < 16; r++) { len[r] = r < rowlen ? ((r == rowlen - 1 ? nnz : ptr[r + 1]) - ptr[r]) : 0; wmax = wmax > len[r] ? wmax : len[r]; }
    int best_w = 0, best_rem = nnz; long long best = entry_b * nnz;
    for (int w = 1; w <= wmax; w++) {
        int rem = 0;
        for (int r = 0; r < 16; r++) rem += len[r] - w > 0 ? len[r] - w : 0;
        const long long b = unit_b * w + entry_b * rem;
        if (b < best) { best = b; best_w = w; best_rem = rem; }
    }
    *remainder = best_rem;
    return best_w;
}

// ---- absorbed list entries (round 6; classic plans).  A COO tile next to an ELL tile of the same tile-row often holds the corner entries of a band that leaves the ELL tile by one column
// (stencils: row 15 -> first column of the next block, row 0 -> last column of the previous one), and the ELL tile's units have padding in exactly those rows.  Such an entry moves
// into the padding slot: the unit's 16-column window of x then starts `shift` columns beside its block (x index = block * 16 + shift + nibble, shift in [-3, 3], kept in flag bits 5-7 of
// the descriptor's word 0), the entry leaves the strip's list.  On the 5-point 4096^2 grid every list entry goes that way: no entry phase, and the lines of x the lists touched long
// before / after the units needed them are fetched once (profiles/r06_config4_x_refetch.txt).  The rule is a pure function of the ELL tile and its two neighbours, evaluated by
// whoever needs it (the ELL tile's emission, the COO tiles' counts and emission), so host and device builders agree by construction.
constexpr int ABSORB_MAX = 24;                               // entries one ELL tile takes at most
constexpr signed char ABSORB_EMPTY = -128;
struct AbsorbPlan {
    int n;                                                   // entries taken
    unsigned char src[ABSORB_MAX], q[ABSORB_MAX];            // from the left (0) / right (1) neighbour, its q-th entry
    signed char col[16][16];                                 // [unit][row]: column relative to the host tile's first column (-3 .. 18), ABSORB_EMPTY = padding
    unsigned char from[16][16];                              // [unit][row]: where the slot's value comes from: 0 .. 15 = that ELL slot of the same row, 0x80 | k = taken entry k
    signed char shift[16], lo[16];                           // per unit: window shift; lowest relative column among its slots (what padding slots point at)
    unsigned short touched;                                  // bit s: unit s differs from the ELL tile's own slot s
    unsigned short derived;                                  // bit s: unit s takes its x from unit s - 1, one lane up (below); dcb / dnib: what its descriptor then holds
    int dcb[16]; unsigned char dnib[16];
    int w;                                                   // units of the host tile (ELL: its width; CSR tile: its split width)
};
// value of the host tile's own slot `f` of row r (AbsorbPlan::from < 16)
TILESPMV_HD inline val_t absorb_own_value(const Tile_matrix *T, int e, int rowlen, int f, int r)
{
    if (T->Format[e] == TILESPMV_FMT_CSR) return T->Blockcsr_Val[T->csr_offset[e] + (T->Blockcsr_Ptr + T->csrptr_offset[e])[r] + f];
    return T->Blockell_Val[T->ell_offset[e] + f * rowlen + r];
}
// A taken entry joins its row at its sorted place (a row's slots hold ascending columns, as the ELL pack leaves them): an entry from the left neighbour goes first and moves the row's
// own entries one slot up — into the padding every row shorter than the tile's width has at its end —, one from the right neighbour goes behind them.  It is taken if afterwards every
// unit's columns still fit a 16-column window that starts at most 3 columns before / after the block.  (5-point grid: row 0 gains column -1, row 15 column 16, and the three
// units of the diagonal tile become the three diagonals r - 1, r, r + 1 with shifts -1, 0, +1 and the identity pattern.)
// DERIVED units (round 6, second step): in a band the slots of a row hold consecutive columns, so unit s is unit s - 1 moved one column to the right — lane r of unit s needs the x that
// lane r + 1 of unit s - 1 has just gathered.  Where that holds for every real slot of lanes 0 .. 14 (the slot one lane up in the previous unit is real and has the same column), unit s
// does not gather: the kernel rotates the previous unit's x by one lane (DPP) and only lane 15 reads x (the other lanes are masked off: the descriptor holds lane 15's column in every
// nibble, under the shift code UNIT_DERIVED_CODE).  5-point grid: the diagonal tile's three units cost one gather and two one-lane loads
// (measured -1.5 % on config 4; a timing-only probe without the one-lane loads had shown -5.5 %: profiles/r06_absorb_ab.txt).
// The tile whose units take part: an ELL tile (its slots), or — csr_split, the ELL-style split of CSR tiles — a CSR tile (the first w entries of every row; what lies beyond w stays on the
// list as before).  A one-entry-per-row piece of an off-diagonal that a grid line two or three columns longer than a multiple of 16 leaves in a tile of its own is such a CSR tile.
TILESPMV_HD inline bool absorb_host_tile(const Tile_matrix *T, int e, bool csr_split) { return T->Format[e] == TILESPMV_FMT_ELL || (csr_split && T->Format[e] == TILESPMV_FMT_CSR); }
TILESPMV_HD inline void ell_absorb_plan(const Tile_matrix *T, int e, int t_lo, int t_hi, int rowlen, AbsorbPlan *A, bool derive = false, int tilen = 0, bool csr_split = false)
{
    A->n = 0; A->touched = 0; A->derived = 0; A->w = 0;
    for (int s = 0; s < 16; s++) { A->shift[s] = 0; A->lo[s] = 0; }
    if (!absorb_host_tile(T, e, csr_split)) return;
    const bool is_csr = T->Format[e] == TILESPMV_FMT_CSR;
    const int cb = T->tile_columnidx[e], off = is_csr ? T->csr_offset[e] : T->ell_offset[e], stored_e = T->blknnz[e + 1] - T->blknnz[e];
    const unsigned char *cptr = is_csr ? T->Blockcsr_Ptr + T->csrptr_offset[e] : nullptr;
    int w;
    if (is_csr) { int rem; w = csr_split_width(cptr, rowlen, stored_e, &rem); } else w = T->tilewidth[e];
    if (w <= 0 || w > 16) return;
    A->w = w;
    bool any = false;
    for (int side = 0; side < 2; side++) {
        const int tn = e + (side ? 1 : -1);
        if (tn >= t_lo && tn < t_hi && T->Format[tn] == TILESPMV_FMT_COO && T->tile_columnidx[tn] == cb + (side ? 1 : -1)) any = true;
    }
    if (!any && !derive) return;
    // a row's own entries are its first len slots (the ELL pack is left-justified, src/csr2tile.h:452-484); padding = value 0 AND column nibble 0 behind the last slot that is not
    // (an entry of A with value 0 keeps its slot: the compat data val[i] = i % 10 is one tenth zeros)
    for (int r = 0; r < 16; r++) {
        int len = 0, k0 = 0;
        if (r < rowlen) {
            if (is_csr) { k0 = cptr[r]; const int k1 = r == rowlen - 1 ? stored_e : cptr[r + 1]; len = k1 - k0 < w ? k1 - k0 : w; }
            else
                for (int s = 0; s < w; s++)
                    if (T->Blockell_Val[off + s * rowlen + r] != (val_t)0 || nib_at(T->ell_compressedIdx, (long long)off + s * rowlen + r) != 0) len = s + 1;
        }
        for (int s = 0; s < w; s++) {
            A->col[s][r] = s >= len ? ABSORB_EMPTY : (signed char)(is_csr ? nib_at(T->csr_compressedIdx, (long long)off + k0 + s) : nib_at(T->ell_compressedIdx, (long long)off + s * rowlen + r));
            A->from[s][r] = (unsigned char)s;
        }
    }
    for (int side = 0; any && side < 2 && A->n < ABSORB_MAX; side++) {
        const int tn = e + (side ? 1 : -1);
        if (tn < t_lo || tn >= t_hi || T->Format[tn] != TILESPMV_FMT_COO || T->tile_columnidx[tn] != cb + (side ? 1 : -1)) continue;
        const int stored = T->blknnz[tn + 1] - T->blknnz[tn], coff = T->coo_offset[tn];
        for (int q = 0; q < stored && q < 256 && A->n < ABSORB_MAX; q++) {
            const unsigned b = T->coo_compressed_Idx[coff + q];
            const int r = (int)(b >> 4), c = side ? 16 + (int)(b & 15u) : (int)(b & 15u) - 16;
            if (c < ABSORB_SHIFT_MIN || c > 15 + ABSORB_SHIFT_MAX || r >= rowlen) continue;
            // the row's slots in order, the new entry at its sorted place
            int have = 0;
            for (int s = 0; s < w; s++) have += A->col[s][r] != ABSORB_EMPTY;
            if (have + 1 > w) continue;   // (no padding left in this row)
            signed char ncol[16]; unsigned char nfrom[16];
            int len = 0; bool placed = false;
            for (int s = 0; s < w; s++) {
                if (A->col[s][r] == ABSORB_EMPTY) continue;
                if (!placed && c < (int)A->col[s][r]) { ncol[len] = (signed char)c; nfrom[len] = (unsigned char)(0x80 | A->n); len++; placed = true; }
                ncol[len] = A->col[s][r]; nfrom[len] = A->from[s][r]; len++;
            }
            if (!placed) { ncol[len] = (signed char)c; nfrom[len] = (unsigned char)(0x80 | A->n); len++; }
            // would every unit still fit its window?
            bool fits = true;
            for (int s = 0; s < w && fits; s++) {
                int lo = 99, hi = -99;
                for (int rr = 0; rr < 16; rr++) {
                    const int v = rr == r ? (s < len ? (int)ncol[s] : (int)ABSORB_EMPTY) : (int)A->col[s][rr];
                    if (v == (int)ABSORB_EMPTY) continue;
                    lo = v < lo ? v : lo; hi = v > hi ? v : hi;
                }
                if (hi >= lo && (hi - lo > 15 || lo < ABSORB_SHIFT_MIN || hi > 15 + ABSORB_SHIFT_MAX)) fits = false;
            }
            if (!fits) continue;
            for (int s = 0; s < w; s++) {
                const signed char v = s < len ? ncol[s] : ABSORB_EMPTY;
                const unsigned char f = s < len ? nfrom[s] : (unsigned char)s;
                if (v != A->col[s][r] || f != A->from[s][r]) A->touched |= (unsigned short)(1u << s);
                A->col[s][r] = v; A->from[s][r] = f;
            }
            A->src[A->n] = (unsigned char)side; A->q[A->n] = (unsigned char)q; A->n++;
        }
    }
    for (int s = 0; s < w; s++) {
        int lo = 99, hi = -99;
        for (int r = 0; r < 16; r++) { const int v = A->col[s][r]; if (v == (int)ABSORB_EMPTY) continue; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        A->shift[s] = (signed char)(hi < lo ? 0 : lo < 0 ? lo : hi > 15 ? hi - 15 : 0);
        A->lo[s] = (signed char)(hi < lo ? 0 : lo);
    }
    if (!derive) return;
    for (int s = 1; s < w; s++) {
        bool ok = true; int real = 0;
        for (int r = 0; r < 15 && ok; r++)
            if (A->col[s][r] != ABSORB_EMPTY) { real++; if (A->col[s - 1][r + 1] != A->col[s][r]) ok = false; }
        if (!ok || real == 0) continue;
        // lane 15's column, written so that the descriptor's arithmetic (block * 16 + shift code's -4 + nibble) gives it back
        const long long a = (long long)cb * 16 + (A->col[s][15] != ABSORB_EMPTY ? (int)A->col[s][15] : 0) + 4;
        if (a < 4 || (a >> 4) > (long long)tilen - 1) continue;
        A->derived |= (unsigned short)(1u << s); A->touched |= (unsigned short)(1u << s);
        A->dcb[s] = (int)(a >> 4); A->dnib[s] = (unsigned char)(a & 15);
    }
}
// which entries of COO tile t its ELL neighbours take: taken[q] = 1; returns how many
TILESPMV_HD inline int coo_absorbed(const Tile_matrix *T, int t, int t_lo, int t_hi, int rowlen, unsigned char *taken, bool csr_split = false)
{
    const int stored = T->blknnz[t + 1] - T->blknnz[t];
    if (taken) for (int q = 0; q < stored && q < 256; q++) taken[q] = 0;
    int n = 0;
    for (int side = 0; side < 2; side++) {   // side 0: this tile is the LEFT neighbour of ELL tile t + 1; side 1: the RIGHT neighbour of ELL tile t - 1
        const int e = t + (side ? -1 : 1);
        if (e < t_lo || e >= t_hi || !absorb_host_tile(T, e, csr_split) || T->tile_columnidx[e] != T->tile_columnidx[t] + (side ? -1 : 1)) continue;
        AbsorbPlan A;
        ell_absorb_plan(T, e, t_lo, t_hi, rowlen, &A, false, 0, csr_split);
        for (int k = 0; k < A.n; k++)
            if (A.src[k] == (unsigned char)side) { n++; if (taken) taken[A.q[k]] = 1; }
    }
    return n;
}

// What one tile adds to its tile-row's counts.  csr_form: 0 CSR tiles stay whole tiles (their own pass), 1 ELL-style split (w units + list entries), 2 pooled units, 3 wide pooled units
// (windows of POOL_WIDE_WINDOW columns, one byte of column offset per slot: hip_plan.h)
// (the pooled nonzeros — CSR tiles, in-tile COO tiles, HYB remainders — are counted per tile-row by pool_row_count, not here).
struct TileCount { int nunits, ncoo, nheavy, ndense, hval, hidx; };
// absorb: list entries of COO tiles move into the padding of neighbouring ELL units where they fit (above); [t_lo, t_hi) = the tiles of t's tile-row
TILESPMV_HD inline TileCount tile_count(const Tile_matrix *T, int t, int rowlen, int tilen, int colA, bool coo_in_tile, bool dense_mfma, int csr_form, bool absorb = false, int t_lo = 0, int t_hi = 0)
{
    TileCount c{0, 0, 0, 0, 0, 0};
    const bool pooled = csr_form >= 2;
    const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
    switch (fmt) {
    case TILESPMV_FMT_ELL: c.nunits = w; break;
    case TILESPMV_FMT_HYB: c.nunits = w; if (coo_in_tile && !pooled) c.ncoo = stored - w * rowlen; break;
    case TILESPMV_FMT_DNSCOL: c.nunits = T->dnscolptr[t + 1] - T->dnscolptr[t]; break;
    case TILESPMV_FMT_DNS:
        if (dense_mfma) c.ndense = 1;
        else c.nunits = tile_collen(T->tile_columnidx[t], tilen, colA);
        break;
    case TILESPMV_FMT_COO: if (coo_in_tile && !pooled) c.ncoo = stored - (absorb ? coo_absorbed(T, t, t_lo, t_hi, rowlen, nullptr, csr_form == 1) : 0); break;
    case TILESPMV_FMT_CSR:
        if (pooled) break;
        if (csr_form == 1) { int rem; c.nunits = csr_split_width(T->Blockcsr_Ptr + T->csrptr_offset[t], rowlen, stored, &rem); c.ncoo = rem; }
        else { c.nheavy = 1; c.hval = stored; c.hidx = 16 + (stored + 1) / 2; }
        break;
    case TILESPMV_FMT_DNSROW: c.nunits = T->dnsrowptr[t + 1] - T->dnsrowptr[t]; break;  // one row unit per dense row
    }
    return c;
}

// ---- pooled units (hip_plan.h): the nonzeros of a tile-row's CSR tiles, COO tiles and HYB remainders, in column-major order (column, then row; tiles are in
// ascending column-block order and a tile's own nonzeros are bucketed by column nibble, so no comparison sort is needed)
struct PoolEnt { unsigned col; unsigned row; val_t val; };   // global column, row inside the tile-row

// Cuts a column-major run of pooled nonzeros into windows: a window starts at the first nonzero not yet taken and holds the (up to 16) following ones whose column is
// less than `width` (16, or POOL_WIDE_WINDOW in wide plans) above its first column.  `col(i)` = column of nonzero i; `emit(begin, end)` is called once per window.
template <class ColOf, class Emit>
TILESPMV_HD inline void pool_windows(long long n, unsigned width, ColOf col, Emit emit)
{
    long long i = 0;
    while (i < n) {
        const unsigned long long lim = (unsigned long long)col(i) + width;
        long long j = i + 1;
        while (j < n && j - i < 16 && (unsigned long long)col(j) < lim) j++;
        emit(i, j);
        i = j;
    }
}

// The pooled nonzeros of one tile: written to out[0 .. count) in column-major order (a stable bucket pass over the tile's row-major entries).  src(k, r, c, v) -> row, column nibble, value
template <class Src>
TILESPMV_HD inline void pool_tile(int cb, int count, Src src, PoolEnt *out)
{
    int start[17];
    for (int c = 0; c < 17; c++) start[c] = 0;
    for (int k = 0; k < count; k++) { unsigned r, c; val_t v; src(k, r, c, v); start[c + 1]++; }
    for (int c = 0; c < 16; c++) start[c + 1] += start[c];
    for (int k = 0; k < count; k++) { unsigned r, c; val_t v; src(k, r, c, v); out[start[c]++] = PoolEnt{(unsigned)cb * 16u + c, r, v}; }
}

// What ONE tile contributes to its tile-row's pool, column-major, written to out[0 .. returned count): CSR tiles always; COO tiles and HYB remainders when they run in-tile.
// hyb_off: byte offset of every HYB tile in hybIdx (nullptr: the matrix has none)
TILESPMV_HD inline int pool_one_tile(const Tile_matrix *T, int t, int rowlen, bool coo_in_tile, const long long *hyb_off, PoolEnt *out)
{
    const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], cb = T->tile_columnidx[t];
    if (fmt == TILESPMV_FMT_CSR) {
        const int off = T->csr_offset[t];
        const unsigned char *ptr = T->Blockcsr_Ptr + T->csrptr_offset[t];
        const unsigned char *idx = T->csr_compressedIdx; const val_t *val = T->Blockcsr_Val;
        unsigned char rowof[256];   // row of entry k (a CSR tile holds fewer than 192 entries)
        for (int r = 0; r < rowlen; r++) { const int k0 = ptr[r], k1 = (r == rowlen - 1) ? stored : ptr[r + 1]; for (int k = k0; k < k1; k++) rowof[k] = (unsigned char)r; }
        const unsigned char *ro = rowof;
        pool_tile(cb, stored, [=](int k, unsigned &r, unsigned &c, val_t &v) { r = ro[k]; c = (unsigned)nib_at(idx, (long long)off + k); v = val[off + k]; }, out);
        return stored;
    }
    if (fmt == TILESPMV_FMT_COO && coo_in_tile) {
        const int off = T->coo_offset[t];
        const unsigned char *idx = T->coo_compressed_Idx; const val_t *val = T->Blockcoo_Val;
        pool_tile(cb, stored, [=](int k, unsigned &r, unsigned &c, val_t &v) { const unsigned b = idx[off + k]; r = b >> 4; c = b & 15u; v = val[off + k]; }, out);
        return stored;
    }
    if (fmt == TILESPMV_FMT_HYB && coo_in_tile && hyb_off) {
        const int off = T->hyb_offset[t], nell = T->tilewidth[t] * rowlen;
        const unsigned char *src = T->hybIdx + hyb_off[t]; const val_t *val = T->Blockhyb_Val;
        pool_tile(cb, stored - nell, [=](int k, unsigned &r, unsigned &c, val_t &v) { const unsigned b = src[(nell + 1) / 2 + k]; r = b >> 4; c = b & 15u; v = val[off + nell + k]; }, out);
        return stored - nell;
    }
    return 0;
}
// ... and how many entries that is, without writing them (the device builder lays a tile-row's tiles out side by side before it fills them)
TILESPMV_HD inline int pool_one_tile_count(const Tile_matrix *T, int t, int rowlen, bool coo_in_tile, const long long *hyb_off)
{
    const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t];
    if (fmt == TILESPMV_FMT_CSR) return stored;
    if (fmt == TILESPMV_FMT_COO && coo_in_tile) return stored;
    if (fmt == TILESPMV_FMT_HYB && coo_in_tile && hyb_off) return stored - T->tilewidth[t] * rowlen;
    return 0;
}

// Everything tile-row bi pools (hip_plan.h "pooled units"), column-major, into out (room for the tile-row's stored nonzeros); returns how many: its tiles' contributions back to back
TILESPMV_HD inline long long pool_row(const Tile_matrix *T, int bi, int rowlen, bool coo_in_tile, const long long *hyb_off, PoolEnt *out)
{
    long long n = 0;
    for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) n += pool_one_tile(T, t, rowlen, coo_in_tile, hyb_off, out + n);
    return n;
}

// upper bound of what pool_row writes for tile-row bi (its stored nonzeros)
TILESPMV_HD inline long long pool_row_capacity(const Tile_matrix *T, int bi) { return (long long)T->blknnz[T->tile_ptr[bi + 1]] - T->blknnz[T->tile_ptr[bi]]; }

// 16-column segments of x (a 128-byte line in fp64; the rule calibrated on them is kept for fp32, where wide windows lose on the same structures) the gathers of window [b, e) touch
TILESPMV_HD inline int pool_window_lines(const PoolEnt *scratch, long long b, long long e)
{
    int nl = 0;
    unsigned last = ~0u;
    for (long long q = b; q < e; q++) { const unsigned l = scratch[q].col >> 4; if (l != last) { nl++; last = l; } }   // (columns ascend inside a window)
    return nl;
}
// pooled part of a tile-row's counts: windows that are worth a unit, and the nonzeros of the others (list entries)
// (nlines, optional: 16-column segments of x — 128-byte lines in fp64 — the units' gathers touch, summed over the units: what a wide window costs the texture path)
TILESPMV_HD inline void pool_windows_count(const PoolEnt *scratch, long long n, unsigned width, int *nunits, int *ncoo, int *nlines)
{
    int nu = 0, nc = 0, nl = 0;
    pool_windows(n, width, [=](long long i) { return scratch[i].col; }, [&](long long b, long long e) {
        if (e - b >= POOL_MIN_FILL) { nu++; nl += pool_window_lines(scratch, b, e); }
        else nc += (int)(e - b);
    });
    *nunits = nu; *ncoo = nc;
    if (nlines) *nlines = nl;
}
TILESPMV_HD inline void pool_row_count(const Tile_matrix *T, int bi, int rowlen, bool coo_in_tile, const long long *hyb_off, unsigned width, PoolEnt *scratch, int *nunits, int *ncoo, int *nlines = nullptr)
{
    const long long n = pool_row(T, bi, rowlen, coo_in_tile, hyb_off, scratch);
    pool_windows_count(scratch, n, width, nunits, ncoo, nlines);
}

// ---- emission
struct EmitOut {   // where the records go (host staging arrays or device memory); entries of arrays a plan does not have are never written
    uint4 *udesc; uint2 *urow; val_t *uval;              // units: descriptor (w0, nibbles 0-7, w0, nibbles 8-15), pooled plans: row nibbles, 16 values
    uint4 *ucol;                                         // wide pooled plans: 16 column-offset bytes per unit (slot s in byte s); their descriptor's nibbles are the ROW nibbles, urow is unused
    val_t *cval; int *ccol; unsigned char *crow;         // list entries: value, global column, tile-row-in-strip << 4 | row
    int *dcb; val_t *dval;                               // dense tiles for the matrix cores: column block, 256 values in operand order
};
struct EmitPos { long long u, c, dq; };   // next unit, list entry, dense tile

TILESPMV_HD inline unsigned unit_word0(bool pooled, unsigned kr, int cb, unsigned flags)
{
    return pooled ? (((unsigned)cb * 16u) | (kr << POOL_KR_SHIFT)) : ((unsigned)cb | (((kr << UNIT_ROW_SHIFT) | flags) << UNIT_FLAG_SHIFT));
}
// one unit of column block cb: src = rowlen consecutive values of this column; nibs: 16 column nibbles, row 0 in the top nibble
TILESPMV_HD inline uint4 nibbles_to_bytes(unsigned long long nibs)   // 16 nibbles (slot 0 in the top nibble) -> 16 bytes (slot s in byte s)
{
    unsigned w[4] = {0u, 0u, 0u, 0u};
    for (int sl = 0; sl < 16; sl++) w[sl >> 2] |= (unsigned)((nibs >> (60 - 4 * sl)) & 15ull) << (8 * (sl & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
TILESPMV_HD inline void put_unit(const EmitOut &O, EmitPos &p, int csr_form, unsigned kr, int cb, const val_t *src, int rows, unsigned long long nibs)
{
    for (int r = 0; r < rows; r++) O.uval[p.u * 16 + r] = src[r];
    const unsigned w0 = unit_word0(csr_form >= 2, kr, cb, 0u);
    if (csr_form == 3) {   // one row per lane: identity row nibbles; the column nibbles become offset bytes
        O.udesc[p.u] = make_uint4(w0, 0x01234567u, w0, 0x89ABCDEFu);
        O.ucol[p.u] = nibbles_to_bytes(nibs);
    } else O.udesc[p.u] = make_uint4(w0, (unsigned)(nibs >> 32), w0, (unsigned)(nibs & 0xffffffffull));
    p.u++;
}

// Unit s of host tile t as the absorb plan left it (a unit that took entries, whose row slots moved, or that is derived): written at unit index u
TILESPMV_HD inline void emit_touched_unit(const Tile_matrix *T, int t, int rowlen, const AbsorbPlan &A, int s, int cb, unsigned kr, int csr_form, const EmitOut &O, long long u)
{
    // its window starts `shift` columns beside the block; padding slots point at the window's lowest used column (derived units: every lane holds lane 15's column)
    const bool der = (A.derived >> s) & 1u;
    const int sh = der ? 0 : A.shift[s], pad = der ? (int)A.dnib[s] : A.lo[s] - sh;
    unsigned long long nibs = 0;
    val_t vals[16];
    for (int r = 0; r < 16; r++) {
        int nb = pad; val_t v = 0;
        if (A.col[s][r] != ABSORB_EMPTY) {
            nb = der ? (int)A.dnib[s] : (int)A.col[s][r] - sh;
            const unsigned f = A.from[s][r];
            if (f & 0x80u) { const int k = (int)(f & 0x7Fu), tn = t + (A.src[k] ? 1 : -1); v = T->Blockcoo_Val[T->coo_offset[tn] + A.q[k]]; }
            else v = absorb_own_value(T, t, rowlen, (int)f, r);
        }
        vals[r] = v;
        nibs |= (unsigned long long)(nb & 15) << (60 - 4 * r);
    }
    EmitPos q{u, 0, 0};
    put_unit(O, q, csr_form, kr, der ? A.dcb[s] : cb, vals, 16, nibs);
    const unsigned shbits = (der ? UNIT_DERIVED_CODE : ((unsigned)sh & 7u)) << UNIT_SHIFT_SHIFT;
    O.udesc[u].x |= shbits; O.udesc[u].z |= shbits;
}

// Everything one tile emits EXCEPT whole ("heavy") CSR tiles (csr_form 0: host only, hip_plan_stream.hip) and the pooled nonzeros (pool_row_emit).
// kr = the tile-row's place in its strip; the caller sets the end-of-row flag on the row's last unit afterwards (classic plans).
TILESPMV_HD inline void tile_emit(const Tile_matrix *T, int t, int rowlen, int tilen, int colA, bool coo_in_tile, bool dense_mfma, int csr_form, unsigned kr, const long long *hyb_off,
                                  const EmitOut &O, EmitPos &p, bool absorb = false, int t_lo = 0, int t_hi = 0, bool derive = false)
{
    const bool pooled = csr_form >= 2;
    const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
    const int cb = T->tile_columnidx[t], collen = tile_collen(cb, tilen, colA);
    switch (fmt) {
    case TILESPMV_FMT_ELL: {
        const int off = T->ell_offset[t];
        AbsorbPlan A;
        A.n = 0; A.touched = 0; A.derived = 0;
        if (absorb && !pooled) ell_absorb_plan(T, t, t_lo, t_hi, rowlen, &A, derive, tilen, csr_form == 1);
        for (int s = 0; s < w; s++) {
            unsigned long long nibs = 0;
            if (!((A.touched >> s) & 1u)) {
                for (int r = 0; r < rowlen; r++) nibs |= (unsigned long long)nib_at(T->ell_compressedIdx, (long long)off + s * rowlen + r) << (60 - 4 * r);
                put_unit(O, p, csr_form, kr, cb, T->Blockell_Val + off + s * rowlen, rowlen, nibs);
                continue;
            }
            emit_touched_unit(T, t, rowlen, A, s, cb, kr, csr_form, O, p.u);
            p.u++;
        }
        break;
    }
    case TILESPMV_FMT_HYB: {
        if (!hyb_off) break;
        const int off = T->hyb_offset[t], nell = w * rowlen;
        const unsigned char *src = T->hybIdx + hyb_off[t];
        for (int s = 0; s < w; s++) {
            unsigned long long nibs = 0;
            for (int r = 0; r < rowlen; r++) nibs |= (unsigned long long)nib_at(src, s * rowlen + r) << (60 - 4 * r);
            put_unit(O, p, csr_form, kr, cb, T->Blockhyb_Val + off + s * rowlen, rowlen, nibs);
        }
        if (coo_in_tile && !pooled)
            for (int q = 0; q < stored - nell; q++) {
                const unsigned char rcb = src[(nell + 1) / 2 + q];
                O.cval[p.c] = T->Blockhyb_Val[off + nell + q]; O.ccol[p.c] = cb * 16 + (rcb & 15);
                O.crow[p.c] = (unsigned char)((kr << 4) | (rcb >> 4)); p.c++;
            }
        break;
    }
    case TILESPMV_FMT_DNSCOL: {
        const int off = T->dnscol_offset[t], co = T->dnscolptr[t], k = T->dnscolptr[t + 1] - co;
        for (int q = 0; q < k; q++) put_unit(O, p, csr_form, kr, cb, T->Blockdensecol_Val + off + q * rowlen, rowlen, 0x1111111111111111ull * (unsigned)(T->densecolid[co + q] & 15));
        break;
    }
    case TILESPMV_FMT_COO:
        if (coo_in_tile && !pooled) {
            const int off = T->coo_offset[t];
            unsigned char taken[256];
            const int ntaken = absorb ? coo_absorbed(T, t, t_lo, t_hi, rowlen, taken, csr_form == 1) : 0;
            for (int q = 0; q < stored; q++) {
                if (ntaken && q < 256 && taken[q]) continue;   // (in a padding slot of a neighbouring ELL unit)
                const unsigned char rcb = T->coo_compressed_Idx[off + q];
                O.cval[p.c] = T->Blockcoo_Val[off + q]; O.ccol[p.c] = cb * 16 + (rcb & 15);
                O.crow[p.c] = (unsigned char)((kr << 4) | (rcb >> 4)); p.c++;
            }
        }
        break;
    case TILESPMV_FMT_DNS: {
        const int off = T->dns_offset[t];
        if (!dense_mfma) {
            for (int q = 0; q < collen; q++) put_unit(O, p, csr_form, kr, cb, T->Blockdense_Val + off + q * rowlen, rowlen, 0x1111111111111111ull * (unsigned)q);
            break;
        }
        // dense tile for the matrix cores: 256 values, zero padded, in MFMA operand order (dense_slot, hip_plan.h)
        val_t *dst = O.dval + p.dq * 256;
        for (int cc = 0; cc < collen; cc++)
            for (int r = 0; r < rowlen; r++) dst[dense_slot(r, cc)] = T->Blockdense_Val[off + cc * rowlen + r];
        O.dcb[p.dq] = cb;
        p.dq++;
        break;
    }
    case TILESPMV_FMT_DNSROW: {
        const int off = T->dnsrow_offset[t], ro = T->dnsrowptr[t], k = T->dnsrowptr[t + 1] - ro;
        for (int q = 0; q < k; q++) {
            for (int cc = 0; cc < collen; cc++) O.uval[p.u * 16 + cc] = T->Blockdenserow_Val[off + q * collen + cc];
            const unsigned w0 = unit_word0(pooled, kr, cb, UNIT_ROWUNIT);
            const unsigned rid = (unsigned)(T->denserowid[ro + q] & 15);
            if (csr_form == 3) {   // slot s = column s of the dense row: offsets 0 .. 15, one row nibble
                O.udesc[p.u] = make_uint4(w0, 0x11111111u * rid, w0, 0x11111111u * rid);
                O.ucol[p.u] = make_uint4(0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu);
            } else if (pooled) {   // slot s = column s of the dense row: identity column nibbles, one row nibble
                O.udesc[p.u] = make_uint4(w0, 0x01234567u, w0, 0x89ABCDEFu);
                O.urow[p.u] = make_uint2(0x11111111u * rid, 0x11111111u * rid);
            } else O.udesc[p.u] = make_uint4(w0, rid, w0, rid);
            p.u++;
        }
        break;
    }
    case TILESPMV_FMT_CSR:
        if (csr_form == 1) {
            const int off = T->csr_offset[t];
            const unsigned char *ptr = T->Blockcsr_Ptr + T->csrptr_offset[t];
            int rem;
            const int ws = csr_split_width(ptr, rowlen, stored, &rem);
            const long long u0 = p.u;
            for (int sidx = 0; sidx < ws; sidx++) {  // descriptors first (zero nibbles), payload below
                const unsigned w0 = (unsigned)cb | ((kr << UNIT_ROW_SHIFT) << UNIT_FLAG_SHIFT);
                O.udesc[p.u] = make_uint4(w0, 0u, w0, 0u);
                p.u++;
            }
            for (int r = 0; r < rowlen; r++) {
                const int k0 = ptr[r], k1 = (r == rowlen - 1) ? stored : ptr[r + 1];
                for (int kk = k0; kk < k1; kk++) {
                    const int lc = nib_at(T->csr_compressedIdx, (long long)off + kk), sidx = kk - k0;
                    if (sidx < ws) {
                        O.uval[(u0 + sidx) * 16 + r] = T->Blockcsr_Val[off + kk];
                        if (r < 8) O.udesc[u0 + sidx].y |= (unsigned)lc << (28 - 4 * r);
                        else O.udesc[u0 + sidx].w |= (unsigned)lc << (28 - 4 * (r - 8));
                    } else {
                        O.cval[p.c] = T->Blockcsr_Val[off + kk]; O.ccol[p.c] = cb * 16 + lc;
                        O.crow[p.c] = (unsigned char)((kr << 4) | r); p.c++;
                    }
                }
            }
            if (absorb && ws > 0) {   // the units this tile's neighbours' entries went into, and derived units, rewritten (plan_tile_ops.h "absorbed list entries": CSR tiles host them too)
                AbsorbPlan A;
                ell_absorb_plan(T, t, t_lo, t_hi, rowlen, &A, derive, tilen, true);
                for (int sidx = 0; sidx < ws && sidx < A.w; sidx++)
                    if ((A.touched >> sidx) & 1u) emit_touched_unit(T, t, rowlen, A, sidx, cb, kr, csr_form, O, u0 + sidx);
            }
        }
        break;
    }
}

// One window [wb, we) of a tile-row's pool: a unit at position u (slot s = s-th nonzero of the window, in row order) when it is full enough, list entries from position c on otherwise
TILESPMV_HD inline void pool_window_put(const PoolEnt *pool, long long wb, long long we, bool wide, unsigned kr, const EmitOut &O, long long u, long long c)
{
    if (we - wb >= POOL_MIN_FILL) {
        const unsigned base = pool[wb].col;
        unsigned cn[2] = {0u, 0u}, rn[2] = {0u, 0u}, cbytes[4] = {0u, 0u, 0u, 0u};
        // slots in ROW order (stable: columns ascending inside a row): the nonzeros of one row sit in neighbouring lanes, which is what the kernel's
        // two interleaved copies of the slab rely on — neighbouring lanes add into different copies, so two nonzeros of a row never meet in one LDS atomic
        int order[16], cnt[17];
        for (int rr = 0; rr < 17; rr++) cnt[rr] = 0;
        for (long long q = wb; q < we; q++) cnt[pool[q].row + 1]++;
        for (int rr = 0; rr < 16; rr++) cnt[rr + 1] += cnt[rr];
        for (long long q = wb; q < we; q++) order[cnt[pool[q].row]++] = (int)(q - wb);
        for (int sl = 0; sl < (int)(we - wb); sl++) {
            const PoolEnt &pe = pool[wb + order[sl]];
            O.uval[u * 16 + sl] = pe.val;
            if (wide) cbytes[sl >> 2] |= (pe.col - base) << (8 * (sl & 3));
            else cn[sl >> 3] |= (pe.col - base) << (28 - 4 * (sl & 7));
            rn[sl >> 3] |= pe.row << (28 - 4 * (sl & 7));
        }
        const unsigned w0 = base | (kr << POOL_KR_SHIFT);
        if (wide) {   // (descriptor nibbles = row nibbles, column offsets as bytes beside it)
            O.udesc[u] = make_uint4(w0, rn[0], w0, rn[1]);
            O.ucol[u] = make_uint4(cbytes[0], cbytes[1], cbytes[2], cbytes[3]);
        } else {
            O.udesc[u] = make_uint4(w0, cn[0], w0, cn[1]);
            O.urow[u] = make_uint2(rn[0], rn[1]);   // (empty slots: value 0, column offset 0, row 0 — they add 0 * x[base] to row 0 of the tile-row)
        }
    } else
        for (long long q = wb; q < we; q++, c++) {
            O.cval[c] = pool[q].val; O.ccol[c] = (int)pool[q].col;
            O.crow[c] = (unsigned char)((kr << 4) | pool[q].row);
        }
}
// the pooled nonzeros of tile-row bi: windows of 16 (or POOL_WIDE_WINDOW) columns -> units, sparse windows -> list entries
TILESPMV_HD inline void pool_windows_emit(const PoolEnt *pool, long long n, unsigned width, unsigned kr, const EmitOut &O, EmitPos &p)
{
    const bool wide = width > 16u;
    pool_windows(n, width, [=](long long q) { return pool[q].col; }, [&](long long wb, long long we) {
        pool_window_put(pool, wb, we, wide, kr, O, p.u, p.c);
        if (we - wb >= POOL_MIN_FILL) p.u++; else p.c += we - wb;
    });
}
TILESPMV_HD inline void pool_row_emit(const Tile_matrix *T, int bi, int rowlen, bool coo_in_tile, unsigned kr, const long long *hyb_off, unsigned width, PoolEnt *pool, const EmitOut &O, EmitPos &p)
{
    const long long n = pool_row(T, bi, rowlen, coo_in_tile, hyb_off, pool);
    pool_windows_emit(pool, n, width, kr, O, p);
}

// ---- packed entry lists (hip_plan.h ERec): one list already in its final order (by column, ties in list order) -> records and per-chunk column bases.
// Chunk k of the list = its records [64k, 64k + 64); base = column of the chunk's first entry; an entry whose column is 2^(32 - dest_bits) or more above the base closes the
// chunk, which is filled up with null records (value 0, offset 0, destination 0: adds 0 * x[base] to the group's first row).  col(i): column of entry i;
// rec(i, base): entry i becomes the next record; pad(): a null record; chunk(base): a chunk begins.
template <class ColAt, class Rec, class Pad, class Chunk>
TILESPMV_HD inline void pack_chunks(long long n, int dest_bits, ColAt col, Rec rec, Pad pad, Chunk chunk)
{
    const unsigned long long span = 1ull << (32 - dest_bits);
    long long i = 0;
    while (i < n) {
        const unsigned b = col(i);
        chunk(b);
        int k = 0;
        while (i < n && k < ECHUNK && (unsigned long long)col(i) - b < span) { rec(i, b); i++; k++; }
        if (i < n) for (; k < ECHUNK; k++) pad();   // interior chunks are always full
    }
}

TILESPMV_HD inline ERec make_erec(val_t v, unsigned w)
{
    ERec r;
#if defined(TILESPMV_F32)
    __builtin_memcpy(&r.v, &v, 4);
#else
    unsigned b[2]; __builtin_memcpy(b, &v, 8); r.lo = b[0]; r.hi = b[1];
#endif
    r.w = w;
    return r;
}

// Where each column panel (2^panel_shift columns) begins in a PACKED list of nrec records (bases B per chunk): off[0 .. NP], relative to the list's begin (off[0] is the
// caller's).  Records are in column order except that the null padding of a chunk closed early repeats the chunk's first column — padding counts as part of the panel of
// the record before it (it adds 0 * x[.] to the group's first row whichever pass executes it).
TILESPMV_HD inline bool erec_is_null(const ERec &rr)   // all bits zero: a padding record, or an entry with value +0, offset 0 and destination 0 (which adds nothing either)
{
#if defined(TILESPMV_F32)
    return rr.w == 0u && rr.v == 0u;
#else
    return rr.w == 0u && rr.lo == 0u && rr.hi == 0u;
#endif
}
TILESPMV_HD inline void panel_offsets(const ERec *R, long long nrec, const unsigned *B, int dest_bits, int panel_shift, int NP, int *off)
{
    unsigned cur = 0;   // panel of the previous record
    int nextp = 1;
    for (long long i = 0; i < nrec; i++) {
        const ERec &rr = R[i];
        const bool null_like = erec_is_null(rr);
        const unsigned here = (B[i / ECHUNK] + (rr.w >> dest_bits)) >> panel_shift;
        const unsigned pnl = null_like && i % ECHUNK != 0 ? cur : (cur > here ? cur : here);
        while (nextp <= (int)pnl) off[nextp++] = (int)i;
        cur = pnl;
    }
    while (nextp <= NP) off[nextp++] = (int)nrec;
}

}  // namespace tilespmv
