// tile_select.h — the per-tile format selection of Tile_create (reference src/csr2tile.h:143-325), written once for the host builder
// (host_tile_create.cpp) and the device builder (hip_tile_create.hip): both call THIS function, so that a tile gets the same format from either.
// Floating point: the row-length variation is computed with separately rounded IEEE double operations in both compilations (contraction is
// switched off in the function: a fused multiply-add in the variance would round differently from the host's and could flip a tile that sits
// exactly on the 0.2 / 1.0 thresholds); double division and square root are correctly rounded on gfx950 as on the host.
#pragma once
#include <cmath>

#include "host_util.h"

namespace tilespmv {

struct Choice { int fmt = 0, stored = 0, width = 0, ndr = 0, ndc = 0, hybcoo = 0, extracted = 0, csrptr = 0; };

// One tile of rowlen x collen.  cnt(r): entries in local row r (0 .. 15).  lrc(k): (row << 4 | column) byte of entry k, k = 0 .. nnz - 1 (read only by the
// whole-column test).  The reference's rule (src/csr2tile.h:150-323).
template <class CntRow, class LrcAt>
TILESPMV_HD inline Choice select_format_reference(int nnz, int rowlen, int collen, CntRow cnt, LrcAt lrc, bool allow_hyb)
{
#pragma clang fp contract(off)
    Choice c;
    if (nnz >= (int)(rowlen * collen * 0.75)) {  // near-dense tile stored dense (src/csr2tile.h:150-158)
        c.fmt = TILESPMV_FMT_DNS; c.stored = rowlen * collen; return c;
    }
    if (nnz <= TILESPMV_COO_NNZ_TH) {  // very sparse tile: COO + copy into the extracted matrix (:159-168)
        c.fmt = TILESPMV_FMT_COO; c.stored = nnz; c.extracted = nnz; return c;
    }
    if (nnz % collen == 0 || nnz % rowlen == 0) {  // candidates for whole-row / whole-column storage (:169-242)
        bool usable = false; int full = 0;
        for (int r = 0; r < rowlen; r++) {
            if (cnt(r) % collen) { usable = false; break; }
            if (cnt(r) == collen) { usable = true; full++; }
        }
        if (usable) { c.fmt = TILESPMV_FMT_DNSROW; c.ndr = full; c.stored = full * collen; return c; }
        unsigned long long ccnt[2] = {0ull, 0ull};   // sixteen 8-bit column counters (the host's uint8_t cnt_col[16]; wraps like it)
        for (int k = 0; k < nnz; k++) { const int j = lrc(k) & 15; const unsigned long long one = 1ull << (8 * (j & 7)), m = 255ull << (8 * (j & 7)); ccnt[j >> 3] = (ccnt[j >> 3] & ~m) | ((ccnt[j >> 3] + one) & m); }
        usable = false; full = 0;
        for (int j = 0; j < collen; j++) {
            const int cj = (int)((ccnt[j >> 3] >> (8 * (j & 7))) & 255ull);
            if (cj % rowlen) { usable = false; break; }
            if (cj == rowlen) { usable = true; full++; }
        }
        if (usable) { c.fmt = TILESPMV_FMT_DNSCOL; c.ndc = full; c.stored = full * rowlen; return c; }
    }
    int widest = 0;
    for (int r = 0; r < rowlen; r++) widest = widest > cnt(r) ? widest : cnt(r);
    const double mean = ((double)nnz) / rowlen;
    double var = 0.0;
    for (int r = 0; r < rowlen; r++) { double d = (double)(cnt(r) - mean); var += d * d; }
    var /= rowlen;
    const double variation = sqrt(var) / mean;  // (:251-265)
    if (variation <= 0.2) {  // regular rows: ELL padded to the widest row (:270-276)
        c.fmt = TILESPMV_FMT_ELL; c.width = widest; c.stored = widest * rowlen; return c;
    }
    if (allow_hyb && variation >= 1.0) {  // dormant in the shipped reference (:279-316)
        const int sv = (int)sizeof(MAT_VAL_TYPE);
        int hw = widest, best = widest * rowlen * sv + (widest * rowlen + 1) / 2, best_spill = 0;
        for (int w = widest - 1; w > 0; w--) {
            int spill = 0;
            for (int r = 0; r < rowlen; r++) spill += cnt(r) - w > 0 ? cnt(r) - w : 0;
            const int b = w * rowlen * sv + (w * rowlen + 1) / 2 + spill * (sv + 1);
            if (best <= b) { hw = w + 1; break; }
            hw = w; best = b; best_spill = spill;
        }
        if (best_spill <= 4) {
            c.fmt = TILESPMV_FMT_HYB; c.width = hw; c.hybcoo = best_spill;
            c.stored = best_spill + hw * rowlen; c.extracted = best_spill; return c;
        }
    }
    c.fmt = TILESPMV_FMT_CSR; c.stored = nnz; c.csrptr = rowlen;  // (:318-323)
    return c;
}

// TILESPMV_CREATE_CDNA4 (opt-in; SURVEY S8 f3, selection side): the reference's thresholds (dense at 75 % fill, COO up to COO_NNZ_TH entries, ELL at a row-length
// variation of 0.2: src/csr2tile.h:150,159,267-270) were tuned for the byte costs of its 32-lane kernels.  Here every format runs as 16-value units (12-byte descriptor
// + 16 values) plus 13- / 9-byte entries, so the choice is made by those bytes: w = the unit width that minimises units + remainder entries; w = 0 -> COO (while it fits
// the reference's COO tile), w = widest row -> ELL, in between CSR (which the plan splits at that very w); dense when 16 whole columns are cheaper than that.
// Whole-row / whole-column tiles keep the reference rule (exact patterns only).
template <class CntRow, class LrcAt>
TILESPMV_HD inline Choice select_format(int nnz, int rowlen, int collen, CntRow cnt, LrcAt lrc, bool allow_hyb, bool cdna4)
{
    if (!cdna4) return select_format_reference(nnz, rowlen, collen, cnt, lrc, allow_hyb);
    Choice c;
    const int sv = (int)sizeof(MAT_VAL_TYPE);
    const long long unit_b = 12 + 16LL * sv, entry_b = sv + 5;
    int widest = 0;
    for (int r = 0; r < rowlen; r++) widest = widest > cnt(r) ? widest : cnt(r);
    int best_w = 0; long long best = entry_b * nnz;
    for (int w = 1; w <= widest; w++) {
        int rem = 0;
        for (int r = 0; r < rowlen; r++) rem += cnt(r) - w > 0 ? cnt(r) - w : 0;
        const long long b = unit_b * w + entry_b * rem;
        if (b < best) { best = b; best_w = w; }
    }
    if ((long long)collen * unit_b <= best) { c.fmt = TILESPMV_FMT_DNS; c.stored = rowlen * collen; return c; }
    if (nnz % collen == 0 || nnz % rowlen == 0) {
        const Choice r = select_format_reference(nnz, rowlen, collen, cnt, lrc, false);
        if (r.fmt == TILESPMV_FMT_DNSROW || r.fmt == TILESPMV_FMT_DNSCOL) return r;
    }
    if (best_w == 0 && nnz <= TILESPMV_COO_NNZ_TH) { c.fmt = TILESPMV_FMT_COO; c.stored = nnz; c.extracted = nnz; return c; }
    if (best_w == widest && widest > 0) { c.fmt = TILESPMV_FMT_ELL; c.width = widest; c.stored = widest * rowlen; return c; }
    c.fmt = TILESPMV_FMT_CSR; c.stored = nnz; c.csrptr = rowlen;
    return c;
}

}  // namespace tilespmv
