// host_tile_create.cpp — CSR -> Tile_matrix on the host (the kept preprocessing API).
//
// Replaces Tile_create / convert_step1..4 of the reference (src/csr2tile.h:5-1020) with an
// O(nnz) many-core algorithm that produces the SAME bytes:
//   * the reference clears and scans tilen-sized scratch per tile-row (O(tilem*tilen),
//     src/csr2tile.h:26,67-71,89) and searches the tile list per nonzero (:406-418); here every
//     tile-row is handled with a stamp array (no clearing) and one stable bucket pass;
//   * all sizes are accumulated in 64 bits and checked against the int32 offsets that the
//     Tile_matrix API exposes (the reference overflows silently, e.g. :912);
//   * threads come from std::thread (no OpenMP runtime dependency).
// What each output field means: SURVEY.md Appendix A.  Selection rules: src/csr2tile.h:143-325.
#include <sys/time.h>
#include <cmath>

#include "host_util.h"
#include "tile_select.h"

namespace tilespmv {
namespace {

struct RowScratch {
    std::vector<int> stamp, local, touched, cursor;
    explicit RowScratch(int tilen) : stamp((size_t)tilen, -1), local((size_t)tilen, 0) {}
};

void pack_nibble_stream(const uint8_t *src, uint8_t *dst, int64_t n)
{
    parallel_chunks((n + 1) / 2, 1 << 16, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            uint8_t hi = src[2 * i], lo = (2 * i + 1 < n) ? src[2 * i + 1] : 0;
            dst[i] = (uint8_t)((hi << 4) + lo);
        }
    });
}

}  // namespace

// First-element-pivot partition sort of the reference (src/utils.h:103-137), restated so that
// rows holding duplicate column ids come out in the same (unstable) order.
void pivot_sort(int *key, val_t *val, int n)
{
    while (n > 1) {
        const int pivot = key[0];
        std::swap(key[0], key[n - 1]); std::swap(val[0], val[n - 1]);
        int lo = 0;
        for (int i = 0; i < n; i++)
            if (key[i] < pivot) { std::swap(key[i], key[lo]); std::swap(val[i], val[lo]); lo++; }
        std::swap(key[n - 1], key[lo]); std::swap(val[n - 1], val[lo]);
        pivot_sort(key, val, lo);
        key += lo + 1; val += lo + 1; n -= lo + 1;
    }
}

void tile_create_impl(Tile_matrix *T, int rowA, int colA, const MAT_PTR_TYPE *rowptr, const int *colidx,
                      const val_t *vals, unsigned flags)
{
    memset(T, 0, sizeof(*T));
    const bool tverbose = getenv("TILESPMV_CREATE_VERBOSE") != nullptr;   // per-phase milliseconds on stderr
    auto now_ms = [] { timeval t; gettimeofday(&t, NULL); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; };
    double tprev = now_ms();
    auto lap = [&](const char *what) { if (!tverbose) return; const double now = now_ms(); fprintf(stderr, "tilespmv: Tile_create %s %.1f ms\n", what, now - tprev); tprev = now; };
    const bool allow_hyb = flags & TILESPMV_CREATE_HYB, cdna4 = flags & TILESPMV_CREATE_CDNA4;
    const int tilem = (rowA + BS - 1) / BS, tilen = (colA + BS - 1) / BS;
    T->tilem = tilem; T->tilen = tilen;
    T->tile_ptr = zalloc<int>((size_t)tilem + 1);
    const int nthreads = host_threads();
    std::vector<RowScratch *> scratch((size_t)nthreads, nullptr);
    auto get_scratch = [&](int tid) { if (!scratch[tid]) scratch[tid] = new RowScratch(tilen); return scratch[tid]; };

    // ---- pass 1: number of populated column blocks per tile-row
    parallel_chunks(tilem, 512, [&](int64_t b, int64_t e, int tid) {
        RowScratch *S = get_scratch(tid);
        for (int bi = (int)b; bi < (int)e; bi++) {
            const int r0 = bi * BS, r1 = std::min(rowA, r0 + BS);
            int n = 0;
            for (int j = rowptr[r0]; j < rowptr[r1]; j++) {
                int cb = colidx[j] >> 4;
                if (S->stamp[cb] != bi) { S->stamp[cb] = bi; n++; }
            }
            T->tile_ptr[bi] = n;
        }
    });
    exclusive_scan_checked(T->tile_ptr, (int64_t)tilem + 1, "tile count");
    const int tilenum = T->tile_ptr[tilem];
    T->tilenum = tilenum;
    if (!(flags & TILESPMV_CREATE_QUIET)) printf("\n  The number of tile = %i\n", tilenum);

    const int64_t nnz_used = rowptr[rowA];
    const size_t np1 = (size_t)tilenum + 1;
    T->tile_columnidx = zalloc<int>(tilenum);
    T->tile_nnz = zalloc<int>(np1);
    uint8_t *cnt_row = zalloc<uint8_t>((size_t)tilenum * BS);
    int *ent = zalloc<int>((size_t)nnz_used);        // CSR position of each entry, tile order
    uint8_t *lrc = zalloc<uint8_t>((size_t)nnz_used);  // (local row << 4) | local col, tile order

    lap("pass 1 (tiles per tile-row)");
    // ---- pass 2: tile list (ascending column block), per-row counts and the tile-ordered gather.
    // Because tiles are numbered tile-row-major, the nonzeros of tile-row bi occupy the same
    // index range [rowptr[16bi], rowptr[16bi+16)) before and after the gather.
    parallel_chunks(tilem, 256, [&](int64_t b, int64_t e, int tid) {
        RowScratch *S = get_scratch(tid);
        for (int bi = (int)b; bi < (int)e; bi++) {
            const int r0 = bi * BS, r1 = std::min(rowA, r0 + BS);
            const int t0 = T->tile_ptr[bi], nt = T->tile_ptr[bi + 1] - t0;
            const int stampv = tilem + bi;  // distinct from pass 1's stamps
            S->touched.clear();
            for (int j = rowptr[r0]; j < rowptr[r1]; j++) {
                int cb = colidx[j] >> 4;
                if (S->stamp[cb] != stampv) { S->stamp[cb] = stampv; S->touched.push_back(cb); }
            }
            if (!std::is_sorted(S->touched.begin(), S->touched.end())) std::sort(S->touched.begin(), S->touched.end());
            S->cursor.assign((size_t)nt + 1, 0);
            for (int k = 0; k < nt; k++) { S->local[S->touched[k]] = k; T->tile_columnidx[t0 + k] = S->touched[k]; }
            for (int r = r0; r < r1; r++)
                for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
                    int k = S->local[colidx[j] >> 4];
                    S->cursor[k + 1]++;
                    cnt_row[(size_t)(t0 + k) * BS + (r - r0)]++;
                }
            int run = rowptr[r0];
            for (int k = 0; k < nt; k++) { int c = S->cursor[k + 1]; T->tile_nnz[t0 + k] = run; S->cursor[k] = run; run += c; }
            for (int r = r0; r < r1; r++)
                for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
                    int k = S->local[colidx[j] >> 4];
                    int pos = S->cursor[k]++;
                    ent[pos] = j;
                    lrc[pos] = (uint8_t)(((r - r0) << 4) | (colidx[j] & 15));
                }
        }
    });
    T->tile_nnz[tilenum] = (int)nnz_used;
    for (auto *s : scratch) delete s;

    lap("pass 2 (tile columns, tile-ordered gather)");
    // ---- per-tile metadata + format selection
    T->Format = zalloc<char>(tilenum);
    T->blknnz = zalloc<int>(np1);
    T->blknnznnz = zalloc<unsigned char>(np1);
    T->dnsrowptr = zalloc<int>(np1);
    T->dnscolptr = zalloc<int>(np1);
    T->tilewidth = zalloc<char>(tilenum);
    T->csr_offset = zalloc<int>(np1);
    T->csrptr_offset = zalloc<int>(np1);
    T->coo_offset = zalloc<int>(np1);
    T->ell_offset = zalloc<int>(np1);
    T->hyb_offset = zalloc<int>(np1);
    T->hyb_coocount = zalloc<int>(np1);
    T->dns_offset = zalloc<int>(np1);
    T->dnsrow_offset = zalloc<int>(np1);
    T->dnscol_offset = zalloc<int>(np1);
    T->new_coocount = zalloc<int>(np1);

    parallel_chunks(tilem, 256, [&](int64_t b, int64_t e, int) {
        for (int bi = (int)b; bi < (int)e; bi++) {
            const int rowlen = tile_rowlen(bi, tilem, rowA);
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                const int collen = tile_collen(T->tile_columnidx[t], tilen, colA);
                const int n = T->tile_nnz[t + 1] - T->tile_nnz[t];
                const uint8_t *cr = cnt_row + (size_t)t * BS, *lr = lrc + T->tile_nnz[t];
                const Choice c = select_format(n, rowlen, collen, [cr](int r) { return (int)cr[r]; }, [lr](int k) { return (int)lr[k]; }, allow_hyb, cdna4);   // (tile_select.h: shared with the device builder)
                T->Format[t] = (char)c.fmt;
                T->blknnz[t] = c.stored;
                T->blknnznnz[t] = (unsigned char)c.stored;
                T->tilewidth[t] = (char)c.width;
                T->dnsrowptr[t] = c.ndr; T->dnscolptr[t] = c.ndc;
                T->hyb_coocount[t] = c.hybcoo; T->new_coocount[t] = c.extracted;
                T->csrptr_offset[t] = c.csrptr;
                int *dst[7] = { T->csr_offset, T->coo_offset, T->ell_offset, T->hyb_offset, T->dns_offset,
                                T->dnsrow_offset, T->dnscol_offset };
                dst[c.fmt][t] = c.stored;
            }
        }
    });
    lap("format selection");
    int64_t hybell = 0;
    {
        std::vector<int64_t> part((size_t)host_threads(), 0);
        parallel_chunks(tilem, 4096, [&](int64_t b, int64_t e, int tid) {
            int64_t acc = 0;
            for (int bi = (int)b; bi < (int)e; bi++) {
                const int rowlen = tile_rowlen(bi, tilem, rowA);
                for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++)
                    if (T->Format[t] == TILESPMV_FMT_HYB) acc += (int64_t)T->tilewidth[t] * rowlen;
            }
            part[(size_t)tid] += acc;
        });
        for (int64_t v : part) hybell += v;
    }
    int *scans[] = { T->csr_offset, T->csrptr_offset, T->coo_offset, T->ell_offset, T->hyb_offset, T->dns_offset,
                     T->dnsrow_offset, T->dnscol_offset, T->dnsrowptr, T->dnscolptr, T->hyb_coocount,
                     T->new_coocount, T->blknnz };
    static const char *names[] = { "csr_offset", "csrptr_offset", "coo_offset", "ell_offset", "hyb_offset", "dns_offset",
                                   "dnsrow_offset", "dnscol_offset", "dnsrowptr", "dnscolptr", "hyb_coocount",
                                   "new_coocount", "blknnz" };
    exclusive_scan_checked_multi(scans, names, (int)(sizeof(scans) / sizeof(*scans)), (int64_t)np1);
    T->csrsize = T->csr_offset[tilenum]; T->csrptrlen = T->csrptr_offset[tilenum];
    T->coosize = T->coo_offset[tilenum]; T->ellsize = T->ell_offset[tilenum];
    T->hybsize = T->hyb_offset[tilenum]; T->hybellsize = (int)hybell; T->hybcoosize = T->hyb_coocount[tilenum];
    T->dnssize = T->dns_offset[tilenum]; T->dnsrowsize = T->dnsrow_offset[tilenum];
    T->dnscolsize = T->dnscol_offset[tilenum]; T->coototal = T->new_coocount[tilenum];

    lap("scans");
    // ---- payload arrays
    T->Blockcsr_Val = zalloc<val_t>(T->csrsize);
    T->Blockcsr_Ptr = zalloc<unsigned char>(T->csrptrlen);
    T->csr_compressedIdx = zalloc<unsigned char>(((size_t)T->csrsize + 1) / 2);
    T->Blockcoo_Val = zalloc<val_t>(T->coosize);
    T->coo_compressed_Idx = zalloc<unsigned char>(T->coosize);
    T->Blockell_Val = zalloc<val_t>(T->ellsize);
    T->ell_compressedIdx = zalloc<unsigned char>(((size_t)T->ellsize + 1) / 2);
    T->Blockhyb_Val = zalloc<val_t>((size_t)T->hybellsize + T->hybcoosize);
    T->hybIdx = zalloc<unsigned char>(((size_t)T->hybellsize + 1) / 2 + T->hybcoosize + (size_t)tilem + 8);
    T->Blockdense_Val = zalloc<val_t>(T->dnssize);
    T->Blockdenserow_Val = zalloc<val_t>(T->dnsrowsize);
    T->denserowid = zalloc<char>(T->dnsrowptr[tilenum]);
    T->Blockdensecol_Val = zalloc<val_t>(T->dnscolsize);
    T->densecolid = zalloc<char>(T->dnscolptr[tilenum]);
    T->deferredcoo_val = zalloc<val_t>(T->coototal);
    T->deferredcoo_colidx = zalloc<int>(T->coototal);
    T->deferredcoo_ptr = zalloc<int>((size_t)rowA + 1);

    uint8_t *csr_col = zalloc<uint8_t>(T->csrsize), *ell_col = zalloc<uint8_t>(T->ellsize);
    uint8_t *hyb_col = zalloc<uint8_t>((size_t)T->hybellsize + T->hybcoosize), *hyb_row = zalloc<uint8_t>(T->hybcoosize);
    int *x_row = zalloc<int>(T->coototal);  // extracted entries in tile order: local row | column | value
    int *x_col = zalloc<int>(T->coototal);
    val_t *x_val = zalloc<val_t>(T->coototal);

    lap("payload allocation");
    // ---- pass 3: pack every tile into its format's arrays (src/csr2tile.h:420-622) and, per
    // tile-row, turn its slice of the extracted list into CSR rows (src/csr2tile.h:899-960).
    parallel_chunks(tilem, 128, [&](int64_t b, int64_t e, int) {
        for (int bi = (int)b; bi < (int)e; bi++) {
            const int rowlen = tile_rowlen(bi, tilem, rowA);
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                const int cb = T->tile_columnidx[t], collen = tile_collen(cb, tilen, colA);
                const int e0 = T->tile_nnz[t], n = T->tile_nnz[t + 1] - e0, w = T->tilewidth[t];
                const uint8_t *rc = lrc + e0; const int *src = ent + e0;
                int start[BS + 1]; start[0] = 0;
                for (int r = 0; r < BS; r++) start[r + 1] = start[r] + cnt_row[(size_t)t * BS + r];
                switch (T->Format[t]) {
                case TILESPMV_FMT_CSR: {
                    const int off = T->csr_offset[t], poff = T->csrptr_offset[t];
                    for (int k = 0; k < n; k++) { T->Blockcsr_Val[off + k] = vals[src[k]]; csr_col[off + k] = rc[k] & 15; }
                    for (int r = 0; r < rowlen; r++) T->Blockcsr_Ptr[poff + r] = (unsigned char)start[r];
                    break;
                }
                case TILESPMV_FMT_COO: {
                    const int off = T->coo_offset[t], xo = T->new_coocount[t];
                    for (int k = 0; k < n; k++) {
                        T->Blockcoo_Val[off + k] = vals[src[k]];
                        T->coo_compressed_Idx[off + k] = rc[k];
                        x_row[xo + k] = rc[k] >> 4; x_col[xo + k] = colidx[src[k]]; x_val[xo + k] = vals[src[k]];
                    }
                    break;
                }
                case TILESPMV_FMT_ELL: {
                    const int off = T->ell_offset[t];
                    for (int k = 0; k < n; k++) {
                        int r = rc[k] >> 4, p = off + (k - start[r]) * rowlen + r;
                        T->Blockell_Val[p] = vals[src[k]]; ell_col[p] = rc[k] & 15;
                    }
                    break;
                }
                case TILESPMV_FMT_HYB: {
                    const int off = T->hyb_offset[t], xo = T->new_coocount[t], ro = T->hyb_coocount[t];
                    int spill = 0;
                    for (int k = 0; k < n; k++) {
                        int r = rc[k] >> 4, s = k - start[r];
                        if (s < w) { T->Blockhyb_Val[off + s * rowlen + r] = vals[src[k]]; hyb_col[off + s * rowlen + r] = rc[k] & 15; }
                        else {
                            T->Blockhyb_Val[off + w * rowlen + spill] = vals[src[k]]; hyb_col[off + w * rowlen + spill] = rc[k] & 15;
                            hyb_row[ro + spill] = (uint8_t)r;
                            x_row[xo + spill] = r; x_col[xo + spill] = colidx[src[k]]; x_val[xo + spill] = vals[src[k]];
                            spill++;
                        }
                    }
                    break;
                }
                case TILESPMV_FMT_DNS: {
                    const int off = T->dns_offset[t];
                    for (int k = 0; k < n; k++) T->Blockdense_Val[off + (rc[k] & 15) * rowlen + (rc[k] >> 4)] = vals[src[k]];
                    break;
                }
                case TILESPMV_FMT_DNSROW: {
                    const int off = T->dnsrow_offset[t], ro = T->dnsrowptr[t];
                    int nr = 0;
                    for (int r = 0; r < rowlen; r++) {
                        if (start[r + 1] - start[r] != collen) continue;
                        T->denserowid[ro + nr++] = (char)r;
                        for (int k = start[r]; k < start[r + 1]; k++) T->Blockdenserow_Val[off + k] = vals[src[k]];
                    }
                    break;
                }
                case TILESPMV_FMT_DNSCOL: {
                    const int off = T->dnscol_offset[t], co = T->dnscolptr[t];
                    for (int k = start[0]; k < start[1]; k++) T->densecolid[co + k] = (char)(rc[k] & 15);
                    for (int k = 0; k < n; k++) { int r = rc[k] >> 4; T->Blockdensecol_Val[off + (k - start[r]) * rowlen + r] = vals[src[k]]; }
                    break;
                }
                }
            }
            // extracted entries of this tile-row -> per-row counts (rows of other tile-rows never appear here)
            const int x0 = T->new_coocount[T->tile_ptr[bi]], x1 = T->new_coocount[T->tile_ptr[bi + 1]];
            for (int i = x0; i < x1; i++) T->deferredcoo_ptr[bi * BS + x_row[i]]++;
        }
    });
    { int *one[] = {T->deferredcoo_ptr}; static const char *nm[] = {"deferredcoo_ptr"}; exclusive_scan_checked_multi(one, nm, 1, (int64_t)rowA + 1); }
    parallel_chunks(tilem, 256, [&](int64_t b, int64_t e, int) {
        for (int bi = (int)b; bi < (int)e; bi++) {
            const int rowlen = tile_rowlen(bi, tilem, rowA);
            const int x0 = T->new_coocount[T->tile_ptr[bi]], x1 = T->new_coocount[T->tile_ptr[bi + 1]];
            if (x0 == x1) continue;
            int fill[BS] = {0};
            for (int i = x0; i < x1; i++) {  // stable scatter in order of appearance (:943-950)
                int r = x_row[i], p = T->deferredcoo_ptr[bi * BS + r] + fill[r]++;
                T->deferredcoo_colidx[p] = x_col[i]; T->deferredcoo_val[p] = x_val[i];
            }
            for (int r = 0; r < rowlen; r++) {
                int p = T->deferredcoo_ptr[bi * BS + r], len = T->deferredcoo_ptr[bi * BS + r + 1] - p;
                int *k = T->deferredcoo_colidx + p;
                bool increasing = true;  // tiles arrive in ascending column order: usually nothing to do
                for (int i = 1; i < len && increasing; i++) increasing = k[i - 1] < k[i];
                if (!increasing) pivot_sort(k, T->deferredcoo_val + p, len);
            }
        }
    });

    // ---- index compression (src/csr2tile.h:973-1008; nibble layout src/encode.h:29-50)
    pack_nibble_stream(csr_col, T->csr_compressedIdx, T->csrsize);
    pack_nibble_stream(ell_col, T->ell_compressedIdx, T->ellsize);
    if (T->hybsize > 0) {
        int64_t src = 0, dst = 0, seen = 0;
        for (int bi = 0; bi < tilem; bi++) {
            const int rowlen = tile_rowlen(bi, tilem, rowA);
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                if (T->Format[t] != TILESPMV_FMT_HYB) continue;
                const int nell = T->tilewidth[t] * rowlen, ncoo = T->blknnz[t + 1] - T->blknnz[t] - nell;
                for (int p = 0; p < nell; p += 2)
                    T->hybIdx[dst + (p >> 1)] = (uint8_t)((hyb_col[src + p] << 4) + (p + 1 < nell ? hyb_col[src + p + 1] : 0));
                dst += (nell + 1) / 2;
                for (int i = 0; i < ncoo; i++) T->hybIdx[dst + i] = (uint8_t)((hyb_row[seen + i] << 4) + hyb_col[src + nell + i]);
                seen += ncoo; src += nell + ncoo; dst += ncoo;
            }
        }
    }
    lap("pass 3 (packing, extraction, nibble streams)");
    free_later({csr_col, ell_col, hyb_col, hyb_row, x_row, x_col, x_val, cnt_row, ent, lrc});   // (0.5 GB for config 4: given back on a detached thread)
    lap("frees");
}

}  // namespace tilespmv

extern "C" {

void Tile_create_ex(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, const MAT_PTR_TYPE *csrRowPtrA,
                    const int *csrColIdxA, const MAT_VAL_TYPE *csrValA, unsigned flags)
{
    (void)nnzA;  // like the reference, the row pointer decides how many nonzeros are used
    tilespmv::tile_create_impl(matrix, rowA, colA, csrRowPtrA, csrColIdxA, csrValA, flags);
}

void Tile_create(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, MAT_PTR_TYPE *csrRowPtrA,
                 int *csrColIdxA, MAT_VAL_TYPE *csrValA)
{
    Tile_create_ex(matrix, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, 0u);
}

void Tile_destroy(Tile_matrix *T)
{
    void *all[] = { T->tile_ptr, T->tile_columnidx, T->tile_nnz, T->Format, T->blknnz, T->blknnznnz, T->dnsrowptr,
        T->dnscolptr, T->tilewidth, T->csr_offset, T->csrptr_offset, T->coo_offset, T->ell_offset, T->hyb_offset,
        T->hyb_coocount, T->dns_offset, T->dnsrow_offset, T->dnscol_offset, T->new_coocount, T->Blockcsr_Val,
        T->Blockcsr_Ptr, T->csr_compressedIdx, T->Blockcoo_Val, T->coo_compressed_Idx, T->Blockell_Val,
        T->ell_compressedIdx, T->Blockhyb_Val, T->hybIdx, T->Blockdense_Val, T->Blockdenserow_Val, T->denserowid,
        T->Blockdensecol_Val, T->densecolid, T->deferredcoo_val, T->deferredcoo_colidx, T->deferredcoo_ptr };
    for (void *p : all) free(p);
    memset(T, 0, sizeof(*T));
}

}  // extern "C"
