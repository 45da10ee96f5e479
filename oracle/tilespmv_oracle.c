/*
 * tilespmv_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A serial, plain-C restatement of the CPU side of SuperScientificSoftwareLaboratory/TileSpMV
 * for the y = A*x path: .mtx -> CSR, CSR -> 16x16 tiles with per-tile format selection and
 * payload packing, the row-block schedule and the serial tile SpMV.  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have something independent
 * to check the HIP path against; nothing under tilespmv_amd/ may link, import or call it.
 *
 * Parity status: PINNED.  Every function here is compared field-by-field / bit-by-bit with the
 * reference's own headers compiled unmodified (oracle/_ref, built by oracle/Makefile from
 * /root/reference/src) in tests/test_oracle_golden.py, and with the known-answer table of
 * SURVEY.md §8(c) (hashes committed under tests/golden/).
 *
 * Each routine cites the reference lines whose behaviour it restates.  The code is organised
 * differently from the reference on purpose (one gather of the tile-ordered entries, then
 * independent per-tile routines) — only the observable outputs are the same.
 *
 * Build: see oracle/Makefile (gcc -O2, -DMAT_VAL_TYPE=double|float).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "../include/tilespmv.h"

#define BS TILESPMV_BLOCK_SIZE
typedef MAT_VAL_TYPE val_t;

#define ORACLE_FLAG_HYB 1u

static void *zalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* In-place exclusive prefix sum over `n` ints (reference src/utils.h:34-48). */
static void excl_scan_int(int *a, int n)
{
    int run = 0;
    for (int i = 0; i < n; i++) { int v = a[i]; a[i] = run; run += v; }
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* The reference sorts each extracted row with a first-element-pivot quicksort
 * (src/utils.h:103-137).  It is not stable, so to be bit-identical even when a row holds
 * duplicate column indices the same partition scheme is restated here. */
static void ref_order_sort(int *key, val_t *val, int n)
{
    while (n > 1) {
        int pivot = key[0];
        int tk = key[0]; key[0] = key[n - 1]; key[n - 1] = tk;
        val_t tv = val[0]; val[0] = val[n - 1]; val[n - 1] = tv;
        int lo = 0;
        for (int i = 0; i < n; i++) {
            if (key[i] < pivot) {
                tk = key[i]; key[i] = key[lo]; key[lo] = tk;
                tv = val[i]; val[i] = val[lo]; val[lo] = tv;
                lo++;
            }
        }
        tk = key[n - 1]; key[n - 1] = key[lo]; key[lo] = tk;
        tv = val[n - 1]; val[n - 1] = val[lo]; val[lo] = tv;
        ref_order_sort(key, val, lo); /* left part recursively, right part by iteration */
        key += lo + 1; val += lo + 1; n -= lo + 1;
    }
}

/* Two 4-bit indices per byte; element p of a stream lives in byte p>>1, high nibble when p is
 * even (reference src/encode.h:29-50). */
static void pack_nibbles(const unsigned char *src, unsigned char *dst, int n)
{
    for (int p = 0; p + 1 < n; p += 2) dst[p >> 1] = (unsigned char)((src[p] << 4) + src[p + 1]);
    if (n & 1) dst[n >> 1] = (unsigned char)(src[n - 1] << 4);
}

static int nibble_at(const unsigned char *packed, int p)
{
    unsigned char b = packed[p >> 1];
    return (p & 1) ? (b & 15) : (b >> 4);
}

/* ------------------------------------------------------------------------------------------
 * Format selection for one tile (reference src/csr2tile.h:143-325).
 * rowcnt[ri] = entries of the tile in local row ri, colcnt[c] = entries in local column c.
 * ---------------------------------------------------------------------------------------- */
typedef struct { int fmt, stored, width, ndr, ndc, hybcoo, extracted, csrptr; } tile_choice;

static tile_choice choose_format(int nnz, int rowlen, int collen, const unsigned char *rowcnt,
                                 const unsigned char *colcnt, unsigned flags)
{
    tile_choice c; memset(&c, 0, sizeof c);
    int dense_threshold = (int)(rowlen * collen * 0.75); /* :150 */
    if (nnz >= dense_threshold) { c.fmt = TILESPMV_FMT_DNS; c.stored = rowlen * collen; return c; }
    if (nnz <= TILESPMV_COO_NNZ_TH) { /* :159-168 */
        c.fmt = TILESPMV_FMT_COO; c.stored = nnz; c.extracted = nnz; return c;
    }
    if (nnz % collen == 0 || nnz % rowlen == 0) { /* :169-242 */
        int ok = 0, full = 0;
        for (int ri = 0; ri < rowlen; ri++) {
            if (rowcnt[ri] % collen != 0) { ok = 0; break; }
            if (rowcnt[ri] == collen) { ok = 1; full++; }
        }
        if (ok) { c.fmt = TILESPMV_FMT_DNSROW; c.ndr = full; c.stored = full * collen; return c; }
        ok = 0; full = 0;
        for (int j = 0; j < collen; j++) {
            if (colcnt[j] % rowlen != 0) { ok = 0; break; }
            if (colcnt[j] == rowlen) { ok = 1; full++; }
        }
        if (ok) { c.fmt = TILESPMV_FMT_DNSCOL; c.ndc = full; c.stored = full * rowlen; return c; }
    }
    /* :245-276 — coefficient of variation of the row lengths decides ELL vs the rest */
    int wmax = 0;
    for (int ri = 0; ri < rowlen; ri++) if (rowcnt[ri] > wmax) wmax = rowcnt[ri];
    double mean = ((double)nnz) / rowlen, var = 0.0;
    for (int ri = 0; ri < rowlen; ri++) { double d = (double)(rowcnt[ri] - mean); var += d * d; }
    var /= rowlen;
    double variation = sqrt(var) / mean;
    if (variation <= 0.2) {
        c.fmt = TILESPMV_FMT_ELL; c.width = wmax; c.stored = wmax * rowlen; return c;
    }
    /* :279-306 — HYB width search: shrink the ELL part while the byte count strictly drops */
    int hw = wmax, coo_best = 0;
    int sv = (int)sizeof(val_t);
    int best = wmax * rowlen * sv + (wmax * rowlen + 1) / 2;
    for (int w = wmax - 1; w > 0; w--) {
        int spill = 0;
        for (int ri = 0; ri < rowlen; ri++) if (rowcnt[ri] > w) spill += rowcnt[ri] - w;
        int bytes = w * rowlen * sv + (w * rowlen + 1) / 2 + spill * (sv + 1);
        if (best <= bytes) { hw = w + 1; break; }
        hw = w; best = bytes; coo_best = spill;
    }
    /* :308-323 — the HYB branch is commented out in the shipped reference */
    if ((flags & ORACLE_FLAG_HYB) && variation >= 1.0 && coo_best <= 4) {
        c.fmt = TILESPMV_FMT_HYB; c.width = hw; c.hybcoo = coo_best;
        c.stored = coo_best + hw * rowlen; c.extracted = coo_best; return c;
    }
    c.fmt = TILESPMV_FMT_CSR; c.stored = nnz; c.csrptr = rowlen;
    return c;
}

/* ------------------------------------------------------------------------------------------
 * CSR -> Tile_matrix (reference src/csr2tile.h:629-1020 and convert_step1..4 :5-627).
 * ---------------------------------------------------------------------------------------- */
void oracle_Tile_create(Tile_matrix *T, int rowA, int colA, MAT_PTR_TYPE nnzA,
                        const MAT_PTR_TYPE *rowptr, const int *colidx, const val_t *vals,
                        unsigned flags)
{
    (void)nnzA;
    memset(T, 0, sizeof *T);
    const int tilem = (rowA + BS - 1) / BS, tilen = (colA + BS - 1) / BS; /* :641-642 */
    T->tilem = tilem; T->tilen = tilen;
    T->tile_ptr = zalloc((size_t)tilem + 1, sizeof(int));

    /* ---- which column blocks are populated in each tile-row (step1 :5-40, step2 :89-101) */
    int *slot = malloc(sizeof(int) * (size_t)(tilen ? tilen : 1)); /* colblock -> local tile */
    for (int i = 0; i < tilen; i++) slot[i] = -1;
    int *touched = malloc(sizeof(int) * (size_t)(tilen ? tilen : 1));
    for (int bi = 0; bi < tilem; bi++) {
        int r0 = bi * BS, r1 = (bi == tilem - 1) ? rowA : r0 + BS, nt = 0;
        for (int j = rowptr[r0]; j < rowptr[r1]; j++) {
            int cb = colidx[j] / BS;
            if (slot[cb] < 0) { slot[cb] = 1; touched[nt++] = cb; }
        }
        for (int k = 0; k < nt; k++) slot[touched[k]] = -1;
        T->tile_ptr[bi] = nt;
    }
    excl_scan_int(T->tile_ptr, tilem + 1); /* :658 */
    const int tilenum = T->tile_ptr[tilem];
    T->tilenum = tilenum;

    T->tile_columnidx = zalloc(tilenum, sizeof(int));
    T->tile_nnz = zalloc((size_t)tilenum + 1, sizeof(int));
    unsigned char *rowcnt = zalloc((size_t)tilenum * BS, 1); /* per tile, per local row */
    const int nnz_used = rowptr[rowA];
    int *ent = zalloc(nnz_used, sizeof(int));             /* CSR position, in tile order */
    unsigned char *ent_row = zalloc(nnz_used, 1);         /* local row of that entry */

    /* ---- tile list per tile-row, ascending column block; counts (step2 :62-102) */
    for (int bi = 0; bi < tilem; bi++) {
        int r0 = bi * BS, r1 = (bi == tilem - 1) ? rowA : r0 + BS, nt = 0;
        int t0 = T->tile_ptr[bi];
        for (int j = rowptr[r0]; j < rowptr[r1]; j++) {
            int cb = colidx[j] / BS;
            if (slot[cb] < 0) { slot[cb] = 1; touched[nt++] = cb; }
        }
        qsort(touched, nt, sizeof(int), cmp_int);
        for (int k = 0; k < nt; k++) { slot[touched[k]] = t0 + k; T->tile_columnidx[t0 + k] = touched[k]; }
        for (int r = r0; r < r1; r++)
            for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
                int t = slot[colidx[j] / BS];
                T->tile_nnz[t]++;
                rowcnt[(size_t)t * BS + (r - r0)]++;
            }
        for (int k = 0; k < nt; k++) slot[touched[k]] = -1;
    }
    excl_scan_int(T->tile_nnz, tilenum + 1); /* :681 */

    /* ---- gather: entries of each tile in row-major order, CSR order inside a row
     *      (what step4's per-tile cursors produce, :403-419) */
    {
        int *cursor = zalloc(tilenum ? tilenum : 1, sizeof(int));
        for (int bi = 0; bi < tilem; bi++) {
            int r0 = bi * BS, r1 = (bi == tilem - 1) ? rowA : r0 + BS;
            int t0 = T->tile_ptr[bi], t1 = T->tile_ptr[bi + 1];
            for (int t = t0; t < t1; t++) slot[T->tile_columnidx[t]] = t;
            for (int r = r0; r < r1; r++)
                for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
                    int t = slot[colidx[j] / BS];
                    int pos = T->tile_nnz[t] + cursor[t]++;
                    ent[pos] = j; ent_row[pos] = (unsigned char)(r - r0);
                }
            for (int t = t0; t < t1; t++) slot[T->tile_columnidx[t]] = -1;
        }
        free(cursor);
    }
    free(slot); free(touched);

    /* ---- per-tile metadata arrays (:683-715) */
    const size_t np1 = (size_t)tilenum + 1;
    T->Format = zalloc(tilenum, 1);
    T->blknnz = zalloc(np1, sizeof(int));
    T->blknnznnz = zalloc(np1, 1);
    T->dnsrowptr = zalloc(np1, sizeof(int));
    T->dnscolptr = zalloc(np1, sizeof(int));
    T->tilewidth = zalloc(tilenum, 1);
    T->csr_offset = zalloc(np1, sizeof(int));
    T->csrptr_offset = zalloc(np1, sizeof(int));
    T->coo_offset = zalloc(np1, sizeof(int));
    T->ell_offset = zalloc(np1, sizeof(int));
    T->hyb_offset = zalloc(np1, sizeof(int));
    T->hyb_coocount = zalloc(np1, sizeof(int));
    T->dns_offset = zalloc(np1, sizeof(int));
    T->dnsrow_offset = zalloc(np1, sizeof(int));
    T->dnscol_offset = zalloc(np1, sizeof(int));
    T->new_coocount = zalloc(np1, sizeof(int));

    /* ---- selection (step3) and size totals (:742-794) */
    for (int bi = 0; bi < tilem; bi++) {
        int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            int collen = (T->tile_columnidx[t] == tilen - 1) ? colA - (tilen - 1) * BS : BS;
            int n = T->tile_nnz[t + 1] - T->tile_nnz[t];
            unsigned char colcnt[BS]; memset(colcnt, 0, sizeof colcnt);
            for (int e = T->tile_nnz[t]; e < T->tile_nnz[t + 1]; e++) colcnt[colidx[ent[e]] % BS]++;
            tile_choice c = choose_format(n, rowlen, collen, rowcnt + (size_t)t * BS, colcnt, flags);
            T->Format[t] = (char)c.fmt;
            T->blknnz[t] = c.stored;
            T->tilewidth[t] = (char)c.width;
            T->dnsrowptr[t] = c.ndr; T->dnscolptr[t] = c.ndc;
            T->hyb_coocount[t] = c.hybcoo; T->new_coocount[t] = c.extracted;
            T->csrptr_offset[t] = c.csrptr;
            switch (c.fmt) {
            case TILESPMV_FMT_CSR: T->csr_offset[t] = c.stored; T->csrsize += c.stored; T->csrptrlen += rowlen; break;
            case TILESPMV_FMT_COO: T->coo_offset[t] = c.stored; T->coosize += c.stored; break;
            case TILESPMV_FMT_ELL: T->ell_offset[t] = c.stored; T->ellsize += c.stored; break;
            case TILESPMV_FMT_HYB: T->hyb_offset[t] = c.stored; T->hybsize += c.stored; T->hybellsize += c.width * rowlen; break;
            case TILESPMV_FMT_DNS: T->dns_offset[t] = c.stored; T->dnssize += c.stored; break;
            case TILESPMV_FMT_DNSROW: T->dnsrow_offset[t] = c.stored; T->dnsrowsize += c.stored; break;
            case TILESPMV_FMT_DNSCOL: T->dnscol_offset[t] = c.stored; T->dnscolsize += c.stored; break;
            }
        }
    }
    for (int t = 0; t <= tilenum; t++) T->blknnznnz[t] = (unsigned char)T->blknnz[t]; /* :796-797 */
    int *scan_these[] = { T->csr_offset, T->csrptr_offset, T->coo_offset, T->ell_offset, T->hyb_offset,
                          T->dns_offset, T->dnsrow_offset, T->dnscol_offset, T->dnsrowptr, T->dnscolptr,
                          T->hyb_coocount, T->new_coocount, T->blknnz }; /* :729-740, :799 */
    for (size_t k = 0; k < sizeof scan_these / sizeof *scan_these; k++) excl_scan_int(scan_these[k], tilenum + 1);
    T->hybcoosize = T->hyb_coocount[tilenum];
    T->coototal = T->new_coocount[tilenum];

    /* ---- payload arrays (:801-869) */
    T->Blockcsr_Val = zalloc(T->csrsize, sizeof(val_t));
    T->Blockcsr_Ptr = zalloc(T->csrptrlen, 1);
    T->csr_compressedIdx = zalloc((T->csrsize + 1) / 2, 1);
    T->Blockcoo_Val = zalloc(T->coosize, sizeof(val_t));
    T->coo_compressed_Idx = zalloc(T->coosize, 1);
    T->Blockell_Val = zalloc(T->ellsize, sizeof(val_t));
    T->ell_compressedIdx = zalloc((T->ellsize + 1) / 2, 1);
    T->Blockhyb_Val = zalloc((size_t)T->hybellsize + T->hybcoosize, sizeof(val_t));
    T->hybIdx = zalloc((size_t)(T->hybellsize + 1) / 2 + T->hybcoosize, 1);
    T->Blockdense_Val = zalloc(T->dnssize, sizeof(val_t));
    T->Blockdenserow_Val = zalloc(T->dnsrowsize, sizeof(val_t));
    T->denserowid = zalloc(T->dnsrowptr[tilenum], 1);
    T->Blockdensecol_Val = zalloc(T->dnscolsize, sizeof(val_t));
    T->densecolid = zalloc(T->dnscolptr[tilenum], 1);

    unsigned char *csr_col = zalloc(T->csrsize, 1), *ell_col = zalloc(T->ellsize, 1);
    unsigned char *hyb_col = zalloc((size_t)T->hybellsize + T->hybcoosize, 1);
    unsigned char *hyb_row = zalloc(T->hybcoosize, 1);
    int *x_row = zalloc(T->coototal, sizeof(int)), *x_col = zalloc(T->coototal, sizeof(int));
    val_t *x_val = zalloc(T->coototal, sizeof(val_t));

    /* ---- packing, one tile at a time (step4 :420-622) */
    for (int bi = 0; bi < tilem; bi++) {
        int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            const int cb = T->tile_columnidx[t];
            const int collen = (cb == tilen - 1) ? colA - (tilen - 1) * BS : BS;
            const int e0 = T->tile_nnz[t], n = T->tile_nnz[t + 1] - e0;
            const int w = T->tilewidth[t];
            int start[BS + 1]; start[0] = 0; /* local row pointer of the tile */
            for (int ri = 0; ri < BS; ri++) start[ri + 1] = start[ri] + rowcnt[(size_t)t * BS + ri];
            switch (T->Format[t]) {
            case TILESPMV_FMT_CSR: { /* :429-451 */
                int off = T->csr_offset[t], poff = T->csrptr_offset[t];
                for (int k = 0; k < n; k++) {
                    T->Blockcsr_Val[off + k] = vals[ent[e0 + k]];
                    csr_col[off + k] = (unsigned char)(colidx[ent[e0 + k]] - cb * BS);
                }
                for (int ri = 0; ri < rowlen; ri++) T->Blockcsr_Ptr[poff + ri] = (unsigned char)start[ri];
                break;
            }
            case TILESPMV_FMT_COO: { /* :452-484, idx byte :975-980 */
                int off = T->coo_offset[t], xo = T->new_coocount[t];
                for (int k = 0; k < n; k++) {
                    int j = ent[e0 + k], lc = colidx[j] - cb * BS, lr = ent_row[e0 + k];
                    T->Blockcoo_Val[off + k] = vals[j];
                    T->coo_compressed_Idx[off + k] = (unsigned char)((lr << 4) + lc);
                    x_row[xo + k] = bi * BS + lr; x_col[xo + k] = colidx[j]; x_val[xo + k] = vals[j];
                }
                break;
            }
            case TILESPMV_FMT_ELL: { /* :485-504 — slot-major, zero padded */
                int off = T->ell_offset[t];
                for (int k = 0; k < n; k++) {
                    int j = ent[e0 + k], lr = ent_row[e0 + k], s = k - start[lr];
                    T->Blockell_Val[off + s * rowlen + lr] = vals[j];
                    ell_col[off + s * rowlen + lr] = (unsigned char)(colidx[j] - cb * BS);
                }
                break;
            }
            case TILESPMV_FMT_HYB: { /* :505-548 — ELL part of width w, remainder as COO + extracted */
                int off = T->hyb_offset[t], xo = T->new_coocount[t], ro = T->hyb_coocount[t], c = 0;
                for (int k = 0; k < n; k++) {
                    int j = ent[e0 + k], lr = ent_row[e0 + k], s = k - start[lr];
                    unsigned char lc = (unsigned char)(colidx[j] - cb * BS);
                    if (s < w) {
                        T->Blockhyb_Val[off + s * rowlen + lr] = vals[j];
                        hyb_col[off + s * rowlen + lr] = lc;
                    } else {
                        T->Blockhyb_Val[off + w * rowlen + c] = vals[j];
                        hyb_col[off + w * rowlen + c] = lc;
                        hyb_row[ro + c] = (unsigned char)lr;
                        x_row[xo + c] = bi * BS + lr; x_col[xo + c] = colidx[j]; x_val[xo + c] = vals[j];
                        c++;
                    }
                }
                break;
            }
            case TILESPMV_FMT_DNS: { /* :549-567 — column-major rowlen x collen */
                int off = T->dns_offset[t];
                for (int k = 0; k < n; k++) {
                    int j = ent[e0 + k];
                    T->Blockdense_Val[off + (colidx[j] - cb * BS) * rowlen + ent_row[e0 + k]] = vals[j];
                }
                break;
            }
            case TILESPMV_FMT_DNSROW: { /* :568-591 — values keep their entry position */
                int off = T->dnsrow_offset[t], ro = T->dnsrowptr[t], nr = 0;
                for (int ri = 0; ri < rowlen; ri++) {
                    if (start[ri + 1] - start[ri] != collen) continue;
                    T->denserowid[ro + nr++] = (char)ri;
                    for (int k = start[ri]; k < start[ri + 1]; k++) T->Blockdenserow_Val[off + k] = vals[ent[e0 + k]];
                }
                break;
            }
            case TILESPMV_FMT_DNSCOL: { /* :592-618 — column ids taken from local row 0 */
                int off = T->dnscol_offset[t], co = T->dnscolptr[t];
                for (int k = start[0]; k < start[1]; k++) T->densecolid[co + k] = (char)(colidx[ent[e0 + k]] - cb * BS);
                for (int k = 0; k < n; k++) {
                    int lr = ent_row[e0 + k], s = k - start[lr];
                    T->Blockdensecol_Val[off + s * rowlen + lr] = vals[ent[e0 + k]];
                }
                break;
            }
            }
        }
    }

    /* ---- extracted very-sparse matrix: COO list -> CSR, rows sorted by column (:899-960) */
    T->deferredcoo_val = zalloc(T->coototal, sizeof(val_t));
    T->deferredcoo_colidx = zalloc(T->coototal, sizeof(int));
    T->deferredcoo_ptr = zalloc((size_t)rowA + 1, sizeof(int));
    for (int i = 0; i < T->coototal; i++) T->deferredcoo_ptr[x_row[i]]++;
    excl_scan_int(T->deferredcoo_ptr, rowA + 1);
    {
        int *fill = zalloc(rowA ? rowA : 1, sizeof(int));
        for (int i = 0; i < T->coototal; i++) {
            int r = x_row[i], p = T->deferredcoo_ptr[r] + fill[r]++;
            T->deferredcoo_val[p] = x_val[i]; T->deferredcoo_colidx[p] = x_col[i];
        }
        free(fill);
    }
    for (int r = 0; r < rowA; r++) {
        int p = T->deferredcoo_ptr[r];
        ref_order_sort(T->deferredcoo_colidx + p, T->deferredcoo_val + p, T->deferredcoo_ptr[r + 1] - p);
    }

    /* ---- index compression (:973-1008) */
    pack_nibbles(csr_col, T->csr_compressedIdx, T->csrsize);
    pack_nibbles(ell_col, T->ell_compressedIdx, T->ellsize);
    {
        int src = 0, dst = 0, rows_seen = 0; /* HYB: per tile a byte-aligned nibble block + row|col bytes */
        for (int bi = 0; bi < tilem; bi++) {
            int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                if (T->Format[t] != TILESPMV_FMT_HYB) continue;
                int nell = T->tilewidth[t] * rowlen, ncoo = T->blknnz[t + 1] - T->blknnz[t] - nell;
                pack_nibbles(hyb_col + src, T->hybIdx + dst, nell);
                dst += (nell + 1) / 2;
                for (int i = 0; i < ncoo; i++)
                    T->hybIdx[dst + i] = (unsigned char)((hyb_row[rows_seen + i] << 4) + hyb_col[src + nell + i]);
                rows_seen += ncoo; src += nell + ncoo; dst += ncoo;
            }
        }
    }
    free(csr_col); free(ell_col); free(hyb_col); free(hyb_row);
    free(x_row); free(x_col); free(x_val);
    free(rowcnt); free(ent); free(ent_row);
}

void oracle_Tile_destroy(Tile_matrix *T)
{
    void *all[] = { T->tile_ptr, T->tile_columnidx, T->tile_nnz, T->Format, T->blknnz, T->blknnznnz,
        T->dnsrowptr, T->dnscolptr, T->tilewidth, T->csr_offset, T->csrptr_offset, T->coo_offset,
        T->ell_offset, T->hyb_offset, T->hyb_coocount, T->dns_offset, T->dnsrow_offset, T->dnscol_offset,
        T->new_coocount, T->Blockcsr_Val, T->Blockcsr_Ptr, T->csr_compressedIdx, T->Blockcoo_Val,
        T->coo_compressed_Idx, T->Blockell_Val, T->ell_compressedIdx, T->Blockhyb_Val, T->hybIdx,
        T->Blockdense_Val, T->Blockdenserow_Val, T->denserowid, T->Blockdensecol_Val, T->densecolid,
        T->deferredcoo_val, T->deferredcoo_colidx, T->deferredcoo_ptr };
    for (size_t i = 0; i < sizeof all / sizeof *all; i++) free(all[i]);
    memset(T, 0, sizeof *T);
}

/* ------------------------------------------------------------------------------------------
 * Row-block schedule (reference src/tilespmv_cpu.h:68-118): tile-rows with more than
 * PREFETCH_SMEM_TH tiles are cut into k = ceil(n/4) chunks of ceil(n/k) tiles.
 * Returns the number of chunks; arrays are malloc'd for the caller.
 * ---------------------------------------------------------------------------------------- */
int oracle_schedule(const Tile_matrix *T, unsigned int **rowidx, int **colstart, int **colstop)
{
    const int TH = TILESPMV_PREFETCH_SMEM_TH;
    int total = 0;
    for (int bi = 0; bi < T->tilem; bi++) {
        int n = T->tile_ptr[bi + 1] - T->tile_ptr[bi];
        total += (n <= TH) ? 1 : (int)ceil((double)n / (double)TH);
    }
    unsigned int *ri = zalloc(total, sizeof *ri);
    int *c0 = zalloc(total, sizeof *c0), *c1 = zalloc(total, sizeof *c1);
    int k = 0;
    for (int bi = 0; bi < T->tilem; bi++) {
        int n = T->tile_ptr[bi + 1] - T->tile_ptr[bi];
        if (n <= TH) { ri[k++] = (unsigned)bi; continue; }
        int parts = (int)ceil((double)n / (double)TH), len = (int)ceil((double)n / (double)parts);
        for (int p = 0; p < parts; p++, k++) {
            ri[k] = (unsigned)bi | 0x80000000u;
            c0[k] = T->tile_ptr[bi] + p * len;
            c1[k] = (p == parts - 1) ? T->tile_ptr[bi] + n : T->tile_ptr[bi] + (p + 1) * len;
        }
    }
    *rowidx = ri; *colstart = c0; *colstop = c1;
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Serial tile SpMV (reference src/tilespmv_cpu.h:125-272).  Tiles in tile order; for every
 * tile a per-row partial sum is formed and then added to y — except dense tiles, which add
 * product by product (:231-235), and COO / HYB-remainder entries (:166, :219).
 * Fills ptroffset1/2 (may be NULL).  Returns the number of rows where y != y_golden
 * (0 if y_golden is NULL) — the reference prints that count (:274-284).
 * ---------------------------------------------------------------------------------------- */
int oracle_tilespmv_cpu(const Tile_matrix *T, int *ptroffset1, int *ptroffset2, int rowA, int colA,
                        const val_t *x, val_t *y, const val_t *y_golden)
{
    const int tilem = T->tilem, tilen = T->tilen;
    int o_csr = 0, o_csrptr = 0, o_coo = 0, o_ell = 0, o_hyb = 0, o_hybidx = 0, o_dns = 0, o_dr = 0, o_dc = 0;
    for (int bi = 0; bi < tilem; bi++) {
        const int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
        val_t *yb = y + (size_t)bi * BS;
        for (int ri = 0; ri < rowlen; ri++) yb[ri] = 0;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            const int collen = (T->tile_columnidx[t] == tilen - 1) ? colA - (tilen - 1) * BS : BS;
            const val_t *xb = x + (size_t)T->tile_columnidx[t] * BS;
            const int stored = T->blknnz[t + 1] - T->blknnz[t];
            const int w = T->tilewidth[t];
            switch (T->Format[t]) {
            case TILESPMV_FMT_CSR:
                if (ptroffset1) { ptroffset1[t] = o_csr; ptroffset2[t] = o_csrptr; }
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    int k1 = (ri == rowlen - 1) ? stored : T->Blockcsr_Ptr[o_csrptr + ri + 1];
                    for (int k = T->Blockcsr_Ptr[o_csrptr + ri]; k < k1; k++)
                        sum += xb[nibble_at(T->csr_compressedIdx, o_csr + k)] * T->Blockcsr_Val[o_csr + k];
                    yb[ri] += sum;
                }
                o_csr += stored; o_csrptr += rowlen;
                break;
            case TILESPMV_FMT_COO:
                if (ptroffset1) ptroffset1[t] = o_coo;
                for (int k = 0; k < stored; k++) {
                    unsigned char b = T->coo_compressed_Idx[o_coo + k];
                    yb[b >> 4] += T->Blockcoo_Val[o_coo + k] * xb[b & 15];
                }
                o_coo += stored;
                break;
            case TILESPMV_FMT_ELL:
                if (ptroffset1) ptroffset1[t] = o_ell;
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int s = 0; s < w; s++) {
                        int p = o_ell + s * rowlen + ri;
                        if (T->Blockell_Val[p] != 0) sum += T->Blockell_Val[p] * xb[nibble_at(T->ell_compressedIdx, p)];
                    }
                    yb[ri] += sum;
                }
                o_ell += w * rowlen;
                break;
            case TILESPMV_FMT_HYB: {
                if (ptroffset1) { ptroffset1[t] = o_hyb; ptroffset2[t] = o_hybidx; }
                const int nell = w * rowlen, ncoo = stored - nell;
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int s = 0; s < w; s++) {
                        int p = s * rowlen + ri;
                        if (T->Blockhyb_Val[o_hyb + p] != 0)
                            sum += T->Blockhyb_Val[o_hyb + p] * xb[nibble_at(T->hybIdx + o_hybidx, p)];
                    }
                    yb[ri] += sum;
                }
                const unsigned char *cb = T->hybIdx + o_hybidx + (nell + 1) / 2;
                for (int i = 0; i < ncoo; i++)
                    yb[cb[i] >> 4] += T->Blockhyb_Val[o_hyb + nell + i] * xb[cb[i] & 15];
                o_hyb += stored; o_hybidx += (nell + 1) / 2 + ncoo;
                break;
            }
            case TILESPMV_FMT_DNS:
                if (ptroffset1) ptroffset1[t] = o_dns;
                for (int ri = 0; ri < rowlen; ri++)
                    for (int c = 0; c < collen; c++)
                        yb[ri] += xb[c] * T->Blockdense_Val[o_dns + c * rowlen + ri];
                o_dns += rowlen * collen;
                break;
            case TILESPMV_FMT_DNSROW:
                if (ptroffset1) ptroffset1[t] = o_dr;
                for (int k = T->dnsrowptr[t]; k < T->dnsrowptr[t + 1]; k++) {
                    val_t sum = 0;
                    for (int c = 0; c < collen; c++)
                        sum += xb[c] * T->Blockdenserow_Val[o_dr + (k - T->dnsrowptr[t]) * collen + c];
                    yb[(int)T->denserowid[k]] += sum;
                }
                o_dr += stored;
                break;
            case TILESPMV_FMT_DNSCOL:
                if (ptroffset1) ptroffset1[t] = o_dc;
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int k = T->dnscolptr[t]; k < T->dnscolptr[t + 1]; k++)
                        sum += T->Blockdensecol_Val[o_dc + (k - T->dnscolptr[t]) * rowlen + ri] * xb[(int)T->densecolid[k]];
                    yb[ri] += sum;
                }
                o_dc += stored;
                break;
            }
        }
    }
    int bad = 0;
    if (y_golden) for (int i = 0; i < rowA; i++) if (y[i] != y_golden[i]) bad++;
    return bad;
}

/* Tile-row-parallel variant of the tile SpMV (OpenMP over tile-rows) for the "all host cores" CPU
 * baseline of SURVEY.md §8(d).  Same per-tile arithmetic and order as oracle_tilespmv_cpu, so y is
 * bit-identical; the per-tile payload offsets are read from the *_offset prefixes that Tile_create
 * stores (they equal the running counters of the serial loop).  HYB tiles need a
 * running byte offset into hybIdx, computed serially first.  Returns the thread count used. */
#ifdef _OPENMP
#include <omp.h>
#endif
static int omp_threads_wanted = 0;   /* 0 = omp_get_max_threads() */
/* Thread count of the all-cores baseline (bench.py passes the cores this process may really use: affinity mask and cgroup
 * quota — omp_get_max_threads() reports every hardware thread of the host, 8x more than a 16-core share allows). */
void oracle_set_omp_threads(int n) { omp_threads_wanted = n > 0 ? n : 0; }

/* Buffer whose pages are first touched by the threads of the baseline, spread over the cores (proc_bind(spread)): a vector
 * allocated and filled by one Python thread sits on one NUMA node and every other node reads it remotely. */
void *oracle_alloc_first_touch(size_t bytes)
{
    char *p = (char *)malloc(bytes ? bytes : 1);
    if (!p) return NULL;
    const long long pages = (long long)((bytes + 4095) / 4096);
#ifdef _OPENMP
    const int nt = omp_threads_wanted > 0 ? omp_threads_wanted : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nt) proc_bind(spread)
#endif
    for (long long i = 0; i < pages; i++) memset(p + (size_t)i * 4096, 0, (size_t)((i + 1) * 4096 <= (long long)bytes ? 4096 : bytes - (size_t)i * 4096));
    return p;
}

int oracle_tilespmv_cpu_omp(const Tile_matrix *T, int rowA, int colA, const val_t *x, val_t *y)
{
    const int tilem = T->tilem, tilen = T->tilen;
    int *hybidx = NULL;
    if (T->hybsize > 0) {
        hybidx = zalloc((size_t)T->tilenum + 1, sizeof(int));
        int at = 0;
        for (int bi = 0; bi < tilem; bi++) {
            const int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++)
                if (T->Format[t] == TILESPMV_FMT_HYB) {
                    const int nell = T->tilewidth[t] * rowlen;
                    hybidx[t] = at; at += (nell + 1) / 2 + (T->blknnz[t + 1] - T->blknnz[t] - nell);
                }
        }
    }
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_threads_wanted > 0 ? omp_threads_wanted : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) proc_bind(spread)
#endif
    for (int bi = 0; bi < tilem; bi++) {
        const int rowlen = (bi == tilem - 1) ? rowA - (tilem - 1) * BS : BS;
        val_t *yb = y + (size_t)bi * BS;
        for (int ri = 0; ri < rowlen; ri++) yb[ri] = 0;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            const int collen = (T->tile_columnidx[t] == tilen - 1) ? colA - (tilen - 1) * BS : BS;
            const val_t *xb = x + (size_t)T->tile_columnidx[t] * BS;
            const int stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
            switch (T->Format[t]) {
            case TILESPMV_FMT_CSR: {
                const int o = T->csr_offset[t], po = T->csrptr_offset[t];
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    int k1 = (ri == rowlen - 1) ? stored : T->Blockcsr_Ptr[po + ri + 1];
                    for (int k = T->Blockcsr_Ptr[po + ri]; k < k1; k++) sum += xb[nibble_at(T->csr_compressedIdx, o + k)] * T->Blockcsr_Val[o + k];
                    yb[ri] += sum;
                }
                break;
            }
            case TILESPMV_FMT_COO: {
                const int o = T->coo_offset[t];
                for (int k = 0; k < stored; k++) { unsigned char b = T->coo_compressed_Idx[o + k]; yb[b >> 4] += T->Blockcoo_Val[o + k] * xb[b & 15]; }
                break;
            }
            case TILESPMV_FMT_ELL: {
                const int o = T->ell_offset[t];
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int s2 = 0; s2 < w; s2++) { int p = o + s2 * rowlen + ri; if (T->Blockell_Val[p] != 0) sum += T->Blockell_Val[p] * xb[nibble_at(T->ell_compressedIdx, p)]; }
                    yb[ri] += sum;
                }
                break;
            }
            case TILESPMV_FMT_HYB: {
                const int o = T->hyb_offset[t], nell = w * rowlen, ncoo = stored - nell;
                const unsigned char *ix = T->hybIdx + hybidx[t];
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int s2 = 0; s2 < w; s2++) { int p = s2 * rowlen + ri; if (T->Blockhyb_Val[o + p] != 0) sum += T->Blockhyb_Val[o + p] * xb[nibble_at(ix, p)]; }
                    yb[ri] += sum;
                }
                ix += (nell + 1) / 2;
                for (int i = 0; i < ncoo; i++) yb[ix[i] >> 4] += T->Blockhyb_Val[o + nell + i] * xb[ix[i] & 15];
                break;
            }
            case TILESPMV_FMT_DNS: {
                const int o = T->dns_offset[t];
                for (int ri = 0; ri < rowlen; ri++) for (int c = 0; c < collen; c++) yb[ri] += xb[c] * T->Blockdense_Val[o + c * rowlen + ri];
                break;
            }
            case TILESPMV_FMT_DNSROW: {
                const int o = T->dnsrow_offset[t];
                for (int k = T->dnsrowptr[t]; k < T->dnsrowptr[t + 1]; k++) {
                    val_t sum = 0;
                    for (int c = 0; c < collen; c++) sum += xb[c] * T->Blockdenserow_Val[o + (k - T->dnsrowptr[t]) * collen + c];
                    yb[(int)T->denserowid[k]] += sum;
                }
                break;
            }
            case TILESPMV_FMT_DNSCOL: {
                const int o = T->dnscol_offset[t];
                for (int ri = 0; ri < rowlen; ri++) {
                    val_t sum = 0;
                    for (int k = T->dnscolptr[t]; k < T->dnscolptr[t + 1]; k++) sum += T->Blockdensecol_Val[o + (k - T->dnscolptr[t]) * rowlen + ri] * xb[(int)T->densecolid[k]];
                    yb[ri] += sum;
                }
                break;
            }
            }
        }
    }
    free(hybidx);
    return nthreads;
}

/* Serial CSR golden (reference src/main.cu:101-110). */
void oracle_csr_spmv(int rowA, const MAT_PTR_TYPE *rowptr, const int *colidx, const val_t *vals,
                     const val_t *x, val_t *y)
{
    for (int i = 0; i < rowA; i++) {
        val_t sum = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; j++) sum += vals[j] * x[colidx[j]];
        y[i] = sum;
    }
}

/* Very-sparse part alone: y += A_coo * x over the extracted CSR (what the reference hands to
 * CSR5 with accumulate semantics: src/tilespmv_cuda.h:1015,1080). */
void oracle_extracted_spmv_add(const Tile_matrix *T, int rowA, const val_t *x, val_t *y)
{
    for (int i = 0; i < rowA; i++) {
        val_t sum = 0;
        for (int j = T->deferredcoo_ptr[i]; j < T->deferredcoo_ptr[i + 1]; j++)
            sum += T->deferredcoo_val[j] * x[T->deferredcoo_colidx[j]];
        y[i] += sum;
    }
}

/* ------------------------------------------------------------------------------------------
 * Matrix Market coordinate reader (reference src/mmio_highlevel.h:593-759 with
 * mm_read_banner / mm_read_mtx_crd_size from src/mmio.h:398-508, :568-603).
 * Return codes as the reference: 0, -1 open, -2 banner, -4 size line.
 * ---------------------------------------------------------------------------------------- */
static void lower(char *s) { for (; *s; s++) *s = (char)tolower((unsigned char)*s); }

int oracle_mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr,
                         int **csrColIdx, val_t **csrVal, const char *filename)
{
    FILE *f = fopen(filename, "r");
    if (!f) return -1;
    char line[1025], banner[65], mtx[65], crd[65], dtype[65], sym[65];
    if (!fgets(line, sizeof line, f) ||
        sscanf(line, "%64s %64s %64s %64s %64s", banner, mtx, crd, dtype, sym) != 5) { fclose(f); return -2; }
    lower(mtx); lower(crd); lower(dtype); lower(sym);
    int is_real = !strcmp(dtype, "real"), is_cplx = !strcmp(dtype, "complex");
    int is_pat = !strcmp(dtype, "pattern"), is_int = !strcmp(dtype, "integer");
    int sym_ok = !strcmp(sym, "general") || !strcmp(sym, "symmetric") || !strcmp(sym, "hermitian") || !strcmp(sym, "skew-symmetric");
    if (strncmp(banner, "%%MatrixMarket", 14) || strcmp(mtx, "matrix") ||
        (strcmp(crd, "coordinate") && strcmp(crd, "array")) ||
        !(is_real || is_cplx || is_pat || is_int) || !sym_ok) { fclose(f); return -2; }
    int mirror = !strcmp(sym, "symmetric") || !strcmp(sym, "hermitian"); /* skew is NOT mirrored (:627) */

    int M = 0, N = 0, NZ = 0;
    do { if (!fgets(line, sizeof line, f)) { fclose(f); return -4; } } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", &M, &N, &NZ) != 3) {
        int got;
        do { got = fscanf(f, "%d %d %d", &M, &N, &NZ); if (got == EOF) { fclose(f); return -4; } } while (got != 3);
    }
    int *cnt = zalloc((size_t)M + 1, sizeof(int));
    int *ri = zalloc(NZ, sizeof(int)), *ci = zalloc(NZ, sizeof(int));
    val_t *vv = zalloc(NZ, sizeof(val_t));
    for (int i = 0; i < NZ; i++) {
        int a = 0, b = 0, iv = 0; double v = 0, im = 0;
        if (is_real) { if (fscanf(f, "%d %d %lg\n", &a, &b, &v) < 0) break; }
        else if (is_cplx) { if (fscanf(f, "%d %d %lg %lg\n", &a, &b, &v, &im) < 0) break; }
        else if (is_int) { if (fscanf(f, "%d %d %d\n", &a, &b, &iv) < 0) break; v = iv; }
        else { if (fscanf(f, "%d %d\n", &a, &b) < 0) break; v = 1.0; }
        a--; b--;
        cnt[a]++; ri[i] = a; ci[i] = b; vv[i] = (val_t)v;
    }
    fclose(f);
    if (mirror) for (int i = 0; i < NZ; i++) if (ri[i] != ci[i]) cnt[ci[i]]++;
    excl_scan_int(cnt, M + 1);
    int total = cnt[M];
    int *rp = zalloc((size_t)M + 1, sizeof(int)); memcpy(rp, cnt, sizeof(int) * ((size_t)M + 1));
    int *cc = zalloc(total, sizeof(int)); val_t *cv = zalloc(total, sizeof(val_t));
    int *fill = zalloc((size_t)M + 1, sizeof(int));
    for (int i = 0; i < NZ; i++) { /* file order; an off-diagonal goes to row i, then to row j (:707-741) */
        int p = rp[ri[i]] + fill[ri[i]]++;
        cc[p] = ci[i]; cv[p] = vv[i];
        if (mirror && ri[i] != ci[i]) {
            p = rp[ci[i]] + fill[ci[i]]++;
            cc[p] = ri[i]; cv[p] = vv[i];
        }
    }
    free(cnt); free(ri); free(ci); free(vv); free(fill);
    *m = M; *n = N; *nnz = total; *isSymmetric = mirror;
    *csrRowPtr = rp; *csrColIdx = cc; *csrVal = cv;
    return 0;
}

void oracle_free(void *p) { free(p); }
int oracle_sizeof_value(void) { return (int)sizeof(val_t); }
