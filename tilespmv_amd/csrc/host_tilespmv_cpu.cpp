// host_tilespmv_cpu.cpp — the kept `tilespmv_cpu` entry point (reference src/tilespmv_cpu.h:3-285):
// row-block schedule + per-tile payload offsets + a serial host SpMV with the reference's
// self-check line.  Part of the drop-in boundary; the GPU path never calls it.
#include <cmath>

#include "host_util.h"

namespace tilespmv {

// Tile-rows with more than PREFETCH_SMEM_TH tiles are cut into k = ceil(n/4) chunks of
// ceil(n/k) tiles, flagged with bit 31 (reference src/tilespmv_cpu.h:68-118).
int build_rowblock_schedule(const Tile_matrix *T, unsigned int **rowidx, int **colstart, int **colstop)
{
    const int TH = TILESPMV_PREFETCH_SMEM_TH;
    auto parts_of = [&](int n) { return n <= TH ? 1 : (n + TH - 1) / TH; };
    int64_t total = 0;
    for (int bi = 0; bi < T->tilem; bi++) total += parts_of(T->tile_ptr[bi + 1] - T->tile_ptr[bi]);
    unsigned int *ri = zalloc<unsigned int>((size_t)total);
    int *c0 = zalloc<int>((size_t)total), *c1 = zalloc<int>((size_t)total);
    int64_t k = 0;
    for (int bi = 0; bi < T->tilem; bi++) {
        const int first = T->tile_ptr[bi], n = T->tile_ptr[bi + 1] - first, parts = parts_of(n);
        if (parts == 1 && n <= TH) { ri[k++] = (unsigned)bi; continue; }
        const int len = (n + parts - 1) / parts;
        for (int p = 0; p < parts; p++, k++) {
            ri[k] = (unsigned)bi | 0x80000000u;
            c0[k] = first + p * len;
            c1[k] = (p == parts - 1) ? first + n : first + (p + 1) * len;
        }
    }
    *rowidx = ri; *colstart = c0; *colstop = c1;
    return (int)total;
}

namespace {
inline int nib(const unsigned char *s, int p) { return (p & 1) ? (s[p >> 1] & 15) : (s[p >> 1] >> 4); }
}  // namespace

}  // namespace tilespmv

using namespace tilespmv;

extern "C" void tilespmv_cpu(Tile_matrix *T, int *ptroffset1, int *ptroffset2, int *rowblkblock,
                             unsigned int **blkcoostylerowidx, int **blkcoostylerowidx_colstart,
                             int **blkcoostylerowidx_colstop, int rowA, int colA, MAT_PTR_TYPE nnzA,
                             MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA, MAT_VAL_TYPE *csrValA, MAT_VAL_TYPE *x,
                             MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden)
{
    (void)nnzA; (void)csrRowPtrA; (void)csrColIdxA; (void)csrValA;
    *rowblkblock = build_rowblock_schedule(T, blkcoostylerowidx, blkcoostylerowidx_colstart, blkcoostylerowidx_colstop);

    // Running element offsets of each format's value stream, in tile order.  They equal the
    // *_offset prefixes of Tile_create (SURVEY.md Appendix A invariants); ptroffset2 is the
    // Blockcsr_Ptr offset (CSR) or the hybIdx byte offset (HYB).
    int at[7] = {0, 0, 0, 0, 0, 0, 0}, csrptr_at = 0, hybidx_at = 0;
    const int tilem = T->tilem, tilen = T->tilen;
    for (int bi = 0; bi < tilem; bi++) {
        const int rowlen = tile_rowlen(bi, tilem, rowA);
        val_t *yb = y + (size_t)bi * BS;
        for (int r = 0; r < rowlen; r++) yb[r] = 0;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
            const int collen = tile_collen(T->tile_columnidx[t], tilen, colA);
            const val_t *xb = x + (size_t)T->tile_columnidx[t] * BS;
            const int off = at[fmt];
            ptroffset1[t] = off;
            val_t acc[BS];
            for (int r = 0; r < BS; r++) acc[r] = 0;
            switch (fmt) {
            case TILESPMV_FMT_CSR: {
                ptroffset2[t] = csrptr_at;
                const unsigned char *ptr = T->Blockcsr_Ptr + csrptr_at;
                for (int r = 0; r < rowlen; r++) {
                    const int k1 = (r == rowlen - 1) ? stored : ptr[r + 1];
                    for (int k = ptr[r]; k < k1; k++) acc[r] += xb[nib(T->csr_compressedIdx, off + k)] * T->Blockcsr_Val[off + k];
                    yb[r] += acc[r];
                }
                csrptr_at += rowlen;
                break;
            }
            case TILESPMV_FMT_COO:
                for (int k = 0; k < stored; k++) {
                    const unsigned char b = T->coo_compressed_Idx[off + k];
                    yb[b >> 4] += T->Blockcoo_Val[off + k] * xb[b & 15];
                }
                break;
            case TILESPMV_FMT_ELL:
                for (int r = 0; r < rowlen; r++) {
                    for (int s = 0; s < w; s++) {
                        const int p = off + s * rowlen + r;
                        if (T->Blockell_Val[p] != 0) acc[r] += T->Blockell_Val[p] * xb[nib(T->ell_compressedIdx, p)];
                    }
                    yb[r] += acc[r];
                }
                break;
            case TILESPMV_FMT_HYB: {
                ptroffset2[t] = hybidx_at;
                const int nell = w * rowlen, ncoo = stored - nell;
                const unsigned char *idx = T->hybIdx + hybidx_at;
                for (int r = 0; r < rowlen; r++) {
                    for (int s = 0; s < w; s++) {
                        const int p = s * rowlen + r;
                        if (T->Blockhyb_Val[off + p] != 0) acc[r] += T->Blockhyb_Val[off + p] * xb[nib(idx, p)];
                    }
                    yb[r] += acc[r];
                }
                idx += (nell + 1) / 2;
                for (int i = 0; i < ncoo; i++) yb[idx[i] >> 4] += T->Blockhyb_Val[off + nell + i] * xb[idx[i] & 15];
                hybidx_at += (nell + 1) / 2 + ncoo;
                break;
            }
            case TILESPMV_FMT_DNS:
                for (int r = 0; r < rowlen; r++)
                    for (int c = 0; c < collen; c++) yb[r] += xb[c] * T->Blockdense_Val[off + c * rowlen + r];
                break;
            case TILESPMV_FMT_DNSROW:
                for (int k = T->dnsrowptr[t]; k < T->dnsrowptr[t + 1]; k++) {
                    val_t s = 0;
                    const val_t *v = T->Blockdenserow_Val + off + (size_t)(k - T->dnsrowptr[t]) * collen;
                    for (int c = 0; c < collen; c++) s += xb[c] * v[c];
                    yb[(int)T->denserowid[k]] += s;
                }
                break;
            case TILESPMV_FMT_DNSCOL:
                for (int r = 0; r < rowlen; r++) {
                    for (int k = T->dnscolptr[t]; k < T->dnscolptr[t + 1]; k++)
                        acc[r] += T->Blockdensecol_Val[off + (k - T->dnscolptr[t]) * rowlen + r] * xb[(int)T->densecolid[k]];
                    yb[r] += acc[r];
                }
                break;
            }
            at[fmt] += stored;
        }
    }
    int errcount = 0;
    if (y_golden)
        for (int i = 0; i < rowA; i++) errcount += (y[i] != y_golden[i]);
    printf(" Run CPU TileSpMV, errcount = %i\n", errcount);
}
