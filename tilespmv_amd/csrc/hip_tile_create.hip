// hip_tile_create.hip — CSR -> Tile_matrix on the device (hip_tile_create.h).
//
// Replaces Tile_create / convert_step1..4 of the reference (src/csr2tile.h:5-1020) a second time: host_tile_create.cpp is the O(nnz) many-core host version, this is the
// device version, producing the SAME bytes (tests/test_gpu_device_build.py compares every member array with the host's).  The shape of the algorithm is not the host's:
//   1. every nonzero gets a 64-bit key  tile-row | column block | local row | local column  (one wavefront per tile-row walks its slice of the CSR arrays, coalesced);
//   2. ONE stable radix sort by (tile-row, column block) puts the nonzeros into tile order — the reference searches the tile list per nonzero (src/csr2tile.h:406-418),
//      the host version buckets per tile-row with a stamp array; the key carries the local coordinates through the sort outside the sorted bit range;
//   3. run-length encoding of the sorted keys = the tile list (tile_ptr, tile_columnidx, tile_nnz);
//   4. one thread per tile: row counts from the keys, the shared selection rule (tile_select.h), sizes -> thirteen exclusive scans;
//   5. one thread per tile packs its payload (values gathered from the CSR array through the sorted positions); nibble streams packed by a second kernel (two tiles may
//      share a byte); the extracted very-sparse matrix (deferredcoo_*) by one more stable sort by row.
// HYB tiles (TILESPMV_CREATE_HYB, dormant in the shipped reference: SURVEY S1; width search src/csr2tile.h:279-306 = tile_select.h, pack :505-548, index bytes :984-1008) are built
// too (round 6): a tile's index bytes are tile-byte-aligned, so the thread that packs the tile writes them itself, at a byte offset that is one more scan.
#include <cstring>

#include <hip/hip_runtime.h>
#include "hip_prims.h"

#include <sys/time.h>
#include <thread>

#include <type_traits>

#include "hip_tile_create.h"
#include "tile_select.h"

namespace tilespmv {
namespace {

#define TC_TRY(expr)                                                                                               \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) {                                                                                    \
            fprintf(stderr, "tilespmv: device Tile_create: HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                                               \
            return -3;                                                                                             \
        }                                                                                                          \
    } while (0)


// the values' upload, by a helper thread on a stream of its own
struct ValueUpload {
    std::thread th;
    hipError_t err = hipSuccess;
    void start(int dev, val_t *dst, const val_t *src, size_t bytes)
    {
        ValueUpload *self = this;
        th = std::thread([=] {
            hipError_t e = hipSetDevice(dev);
            hipStream_t cs = nullptr;
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
            if (e == hipSuccess) {
                e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs);
                if (e == hipSuccess) e = hipStreamSynchronize(cs);
                (void)hipStreamDestroy(cs);
            }
            self->err = e;
        });
    }
    hipError_t wait() { if (th.joinable()) th.join(); return err; }
    ~ValueUpload() { if (th.joinable()) th.join(); }
};


inline double now_ms() { timeval t; gettimeofday(&t, NULL); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }
inline int bits_for(int n) { int b = 1; while (b < 31 && (1ll << b) < (long long)n) b++; return b; }   // bits that hold 0 .. n - 1 (at least one)

typedef unsigned long long u64;

// ---- 1. keys: one wavefront per tile-row
__global__ __launch_bounds__(256) void k_tc_keys(int rowA, int tilem, const int *__restrict__ rowptr, const int *__restrict__ colidx, int bi_shift, u64 *__restrict__ key, int *__restrict__ ent)
{
    __shared__ int s_rp[4][17];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long bi = (long long)blockIdx.x * 4 + wv;
    if (bi < tilem && lane < 17) s_rp[wv][lane] = rowptr[min(16LL * bi + lane, (long long)rowA)];
    __syncthreads();
    if (bi >= tilem) return;
    const int j0 = s_rp[wv][0], j1 = s_rp[wv][16];
    for (int j = j0 + lane; j < j1; j += 64) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < 16; k++) r += s_rp[wv][k] <= j;   // local row of CSR position j
        const unsigned c = (unsigned)colidx[j];
        key[j] = ((u64)bi << bi_shift) | ((u64)(c >> 4) << 8) | (u64)((r << 4) | (int)(c & 15u));
        ent[j] = j;
    }
}

constexpr unsigned KEY_TILE_SHIFT = 8;   // key >> 8 = (tile-row, column block): what names a tile

// ---- 3. the tile list from the run-length encoded keys
__global__ void k_tc_tiles(int tilenum, int tilem, int cb_bits, const u64 *__restrict__ uniq, int *__restrict__ tile_columnidx, int *__restrict__ tile_bi, int *__restrict__ tile_ptr)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tilenum) return;
    const u64 u = uniq[t];
    const int bi = (int)(u >> cb_bits), cb = (int)(u & ((1ull << cb_bits) - 1ull));
    tile_columnidx[t] = cb; tile_bi[t] = bi;
    const int prev = t ? (int)(uniq[t - 1] >> cb_bits) : -1;
    for (int b = prev + 1; b <= bi; b++) tile_ptr[b] = (int)t;   // tile-rows without tiles in between point at this tile too
    if (t == tilenum - 1) for (int b = bi + 1; b <= tilem; b++) tile_ptr[b] = tilenum;
}

struct TileArrays {   // the per-tile arrays of Tile_matrix (device)
    char *Format; int *blknnz; unsigned char *blknnznnz; int *dnsrowptr, *dnscolptr; char *tilewidth;
    int *csrptr_offset, *hyb_coocount, *new_coocount;
    int *fmt_offset[7];   // csr, coo, ell, hyb, dns, dnsrow, dnscol (TILESPMV_FMT_* order)
    int *hyb_bytes;       // HYB tiles: bytes of the tile in hybIdx (scanned into the tile's byte offset)
};

// ---- 4. selection: one thread per tile
__global__ __launch_bounds__(256) void k_tc_select(int tilenum, int tilem, int tilen, int rowA, int colA, bool allow_hyb, bool cdna4, const int *__restrict__ tile_nnz, const int *__restrict__ tile_bi,
                                                     const int *__restrict__ tile_columnidx, const u64 *__restrict__ key, TileArrays A)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tilenum) return;
    const int e0 = tile_nnz[t], n = tile_nnz[t + 1] - e0;
    const int rowlen = tile_rowlen(tile_bi[t], tilem, rowA), collen = tile_collen(tile_columnidx[t], tilen, colA);
    u64 c0 = 0, c1 = 0;   // sixteen 8-bit row counters
    for (int k = 0; k < n; k++) {
        const int r = (int)(key[e0 + k] >> 4) & 15;
        const u64 one = 1ull << (8 * (r & 7)), m = 255ull << (8 * (r & 7));
        if (r < 8) c0 = (c0 & ~m) | ((c0 + one) & m); else c1 = (c1 & ~m) | ((c1 + one) & m);
    }
    const u64 *kp = key + e0;
    const Choice c = select_format(n, rowlen, collen, [c0, c1](int r) { return (int)(((r < 8 ? c0 : c1) >> (8 * (r & 7))) & 255ull); }, [kp](int k) { return (int)(kp[k] & 255ull); }, allow_hyb, cdna4);
    A.Format[t] = (char)c.fmt;
    A.blknnz[t] = c.stored;
    A.blknnznnz[t] = (unsigned char)c.stored;
    A.tilewidth[t] = (char)c.width;
    A.dnsrowptr[t] = c.ndr; A.dnscolptr[t] = c.ndc;
    A.hyb_coocount[t] = c.hybcoo; A.new_coocount[t] = c.extracted;
    A.csrptr_offset[t] = c.csrptr;
    A.fmt_offset[c.fmt][t] = c.stored;
    if (c.fmt == TILESPMV_FMT_HYB) A.hyb_bytes[t] = (c.width * rowlen + 1) / 2 + c.hybcoo;   // ELL part in nibbles (rounded up to a byte), then one byte per remainder entry
}

// totals of K int arrays of n elements each, in 64 bits (the scans below are done in int: a total that fits proves every prefix does, the counts are non-negative)
struct ScanSet { int *a[16]; };
__global__ __launch_bounds__(256) void k_tc_totals(ScanSet S, int K, long long n, unsigned long long *__restrict__ totals)
{
    for (int k = 0; k < K; k++) {
        long long acc = 0;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += S.a[k][i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&totals[k], (unsigned long long)acc);
    }
}

struct PackArrays {
    val_t *Blockcsr_Val; unsigned char *Blockcsr_Ptr, *csr_col;   // csr_col / ell_col: one byte per slot, packed into nibbles afterwards
    val_t *Blockcoo_Val; unsigned char *coo_compressed_Idx;
    val_t *Blockell_Val; unsigned char *ell_col;
    val_t *Blockdense_Val, *Blockdenserow_Val, *Blockdensecol_Val;
    char *denserowid, *densecolid;
    val_t *Blockhyb_Val; unsigned char *hybIdx; const int *hyb_byte_off;   // HYB: values, index bytes, every tile's byte offset in hybIdx
    unsigned *x_key; int *x_col; val_t *x_val;   // extracted entries in tile order: global row, column, value (nullptr: not wanted)
    int *deferredcoo_ptr;                          // per-row counts (atomics), scanned afterwards
};

// ---- 5. packing: one thread per tile (src/csr2tile.h:420-622)
__global__ __launch_bounds__(256) void k_tc_pack(int tilenum, int tilem, int tilen, int rowA, int colA, const Tile_matrix T, const int *__restrict__ tile_bi, const u64 *__restrict__ key,
                                                   const int *__restrict__ ent, const int *__restrict__ colidx, const val_t *__restrict__ vals, PackArrays P)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tilenum) return;
    const int bi = tile_bi[t], cb = T.tile_columnidx[t];
    const int rowlen = tile_rowlen(bi, tilem, rowA);
    (void)cb; (void)tilen; (void)colA;
    const int e0 = T.tile_nnz[t], n = T.tile_nnz[t + 1] - e0, fmt = T.Format[t];
    const u64 *kp = key + e0; const int *src = ent + e0;
    int rcur = 0, rstart = 0;   // the row the walk is in and where it starts (entries of a tile are in row order)
    switch (fmt) {
    case TILESPMV_FMT_CSR: {
        const int off = T.csr_offset[t], poff = T.csrptr_offset[t];
        P.Blockcsr_Ptr[poff] = 0;
        for (int k = 0; k < n; k++) {
            const int rc = (int)(kp[k] & 255ull), r = rc >> 4;
            while (rcur < r) { rcur++; if (rcur < rowlen) P.Blockcsr_Ptr[poff + rcur] = (unsigned char)k; }
            P.Blockcsr_Val[off + k] = vals[src[k]]; P.csr_col[off + k] = (unsigned char)(rc & 15);
        }
        while (rcur < rowlen - 1) { rcur++; P.Blockcsr_Ptr[poff + rcur] = (unsigned char)n; }
        break;
    }
    case TILESPMV_FMT_COO: {
        const int off = T.coo_offset[t], xo = T.new_coocount[t];
        for (int k = 0; k < n; k++) {
            const int rc = (int)(kp[k] & 255ull);
            const val_t v = vals[src[k]];
            P.Blockcoo_Val[off + k] = v;
            P.coo_compressed_Idx[off + k] = (unsigned char)rc;
            if (P.x_key) {
                const unsigned row = (unsigned)bi * 16u + (unsigned)(rc >> 4);
                P.x_key[xo + k] = row; P.x_col[xo + k] = colidx[src[k]]; P.x_val[xo + k] = v;
                atomicAdd(&P.deferredcoo_ptr[row], 1);
            }
        }
        break;
    }
    case TILESPMV_FMT_ELL: {
        const int off = T.ell_offset[t];
        for (int k = 0; k < n; k++) {
            const int rc = (int)(kp[k] & 255ull), r = rc >> 4;
            if (r != rcur) { rcur = r; rstart = k; }
            const int p = off + (k - rstart) * rowlen + r;
            P.Blockell_Val[p] = vals[src[k]]; P.ell_col[p] = (unsigned char)(rc & 15);
        }
        break;
    }
    case TILESPMV_FMT_HYB: {   // ELL part of width w (slot-major, zero padded) + the entries beyond it in row order (src/csr2tile.h:505-548); index bytes :984-1008
        const int off = T.hyb_offset[t], xo = T.new_coocount[t], w = T.tilewidth[t], nell = w * rowlen;
        unsigned char *ib = P.hybIdx + P.hyb_byte_off[t];   // this tile's bytes (nobody else's: the stream is tile-byte-aligned)
        int spill = 0;
        for (int k = 0; k < n; k++) {
            const int rc = (int)(kp[k] & 255ull), r = rc >> 4;
            if (r != rcur) { rcur = r; rstart = k; }
            const int sl = k - rstart;
            const val_t v = vals[src[k]];
            if (sl < w) {
                const int q = sl * rowlen + r;
                P.Blockhyb_Val[off + q] = v;
                ib[q >> 1] = (unsigned char)(ib[q >> 1] | ((q & 1) ? (rc & 15) : ((rc & 15) << 4)));   // nibble at position q of the tile's own stream: high nibble first
            } else {
                P.Blockhyb_Val[off + nell + spill] = v;
                ib[(nell + 1) / 2 + spill] = (unsigned char)rc;   // (row << 4) | column
                if (P.x_key) {
                    const unsigned row = (unsigned)bi * 16u + (unsigned)r;
                    P.x_key[xo + spill] = row; P.x_col[xo + spill] = colidx[src[k]]; P.x_val[xo + spill] = v;
                    atomicAdd(&P.deferredcoo_ptr[row], 1);
                }
                spill++;
            }
        }
        break;
    }
    case TILESPMV_FMT_DNS: {
        const int off = T.dns_offset[t];
        for (int k = 0; k < n; k++) { const int rc = (int)(kp[k] & 255ull); P.Blockdense_Val[off + (rc & 15) * rowlen + (rc >> 4)] = vals[src[k]]; }
        break;
    }
    case TILESPMV_FMT_DNSROW: {   // every row is full or empty: the values of the full rows back to back, their row ids in order
        const int off = T.dnsrow_offset[t], ro = T.dnsrowptr[t];
        int nr = 0, last = -1;
        for (int k = 0; k < n; k++) {
            const int r = (int)(kp[k] >> 4) & 15;
            if (r != last) { P.denserowid[ro + nr++] = (char)r; last = r; }
            P.Blockdenserow_Val[off + k] = vals[src[k]];
        }
        break;
    }
    case TILESPMV_FMT_DNSCOL: {
        const int off = T.dnscol_offset[t], co = T.dnscolptr[t];
        for (int k = 0; k < n; k++) {
            const int rc = (int)(kp[k] & 255ull), r = rc >> 4;
            if (r != rcur) { rcur = r; rstart = k; }
            if (r == 0) P.densecolid[co + k] = (char)(rc & 15);   // the columns present = the columns of row 0
            P.Blockdensecol_Val[off + (k - rstart) * rowlen + r] = vals[src[k]];
        }
        break;
    }
    default: break;
    }
}

// two index bytes -> one byte of a nibble stream (src/encode.h:29-50)
__global__ void k_tc_nibbles(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i >= n) return;
    const unsigned hi = src[2 * i], lo = (2 * i + 1 < n) ? src[2 * i + 1] : 0u;
    dst[i] = (unsigned char)((hi << 4) + lo);
}

// the extracted matrix: entries in their final (row, order of appearance) order through the sorted positions
__global__ void k_tc_deferred(int n, const int *__restrict__ pos, const int *__restrict__ x_col, const val_t *__restrict__ x_val, int *__restrict__ colidx, val_t *__restrict__ val)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    colidx[i] = x_col[pos[i]]; val[i] = x_val[pos[i]];
}
__global__ void k_tc_widen(long long n, const int *__restrict__ a, long long *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i];
}
__global__ void k_tc_iota(int n, int *__restrict__ a)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (int)i;
}
__global__ void k_tc_unsorted_rows(int rowA, const int *__restrict__ ptr, const int *__restrict__ colidx, int *__restrict__ count)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rowA) return;
    for (int i = ptr[r] + 1; i < ptr[r + 1]; i++) if (colidx[i - 1] >= colidx[i]) { atomicAdd(count, 1); return; }
}

// Device arrays come from a few POOLS (one zeroed hipMalloc per phase of the build: the per-tile arrays, the payload arrays) instead of one hipMalloc + memset each — the build
// makes about forty arrays, and on a busy allocator their calls, not the kernels, set the time (one box: 62 ms in the packing phase against 6 ms of kernels).  An array that
// does not fit what is left of the current pool (or is asked for outside one) gets a hipMalloc of its own.
struct Pool { char *at = nullptr; size_t left = 0; };
thread_local Pool t_pool;   // (a build runs on one host thread; two builds on two threads have a pool each)
inline size_t pool_need(size_t n, size_t elem) { return (std::max<size_t>(n, 1) * elem + 16 + 255) / 256 * 256; }
int pool_begin(DevTile *D, size_t bytes)
{
    void *p = nullptr;
    t_pool = Pool();
    TC_TRY(hipMalloc(&p, bytes + 256));
    D->pools.push_back(p);   // (its own list: dfree never looks here — the first array carved from a pool has the pool's address, and freeing IT must not free the pool)
    TC_TRY(hipMemsetAsync(p, 0, bytes + 256, 0));
    t_pool.at = (char *)p; t_pool.left = bytes + 256;
    return 0;
}
inline void pool_end() { t_pool = Pool(); }
template <class V>
int dalloc(DevTile *D, V **out, size_t n, bool zero)
{
    const size_t need = pool_need(n, sizeof(V));
    if (t_pool.left >= need) {   // (pool memory is zeroed already)
        *out = (V *)t_pool.at; t_pool.at += need; t_pool.left -= need;
        return 0;
    }
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(V) + 16;
    TC_TRY(hipMalloc(&p, bytes));
    D->allocs.push_back(p);
    if (zero) TC_TRY(hipMemsetAsync(p, 0, bytes, 0));
    *out = (V *)p;
    return 0;
}
// hipFree waits for the whole device — also for the values' upload that runs beside the first steps of create_impl: while that is under way, memory to be released is only noted
thread_local std::vector<void *> *t_free_later = nullptr;
inline void free_now_or_later(void *p)
{
    if (!p) return;
    if (t_free_later) t_free_later->push_back(p); else (void)hipFree(p);
}
inline void dfree(DevTile *D, const void *p)   // (an array carved from a pool is not in D->allocs: it goes with its pool, at devtile_destroy)
{
    if (!p) return;
    auto it = std::find(D->allocs.begin(), D->allocs.end(), (void *)p);
    if (it == D->allocs.end()) return;
    D->allocs.erase(it);
    free_now_or_later((void *)p);
}
inline unsigned blocks_for(long long n, int per) { return (unsigned)std::max<long long>(1, (n + per - 1) / per); }

int create_impl(DevTile *D, int rowA, int colA, const MAT_PTR_TYPE *h_rowptr, const int *h_colidx, const val_t *h_val, unsigned flags, bool want_deferred, bool csr_on_device)
{
    const bool verbose = getenv("TILESPMV_CREATE_VERBOSE") != nullptr, cdna4 = flags & TILESPMV_CREATE_CDNA4, allow_hyb = flags & TILESPMV_CREATE_HYB;
    pool_end();   // (no pool left over from a build that failed on this thread)
    Tile_matrix &T = D->T;
    const int tilem = (rowA + BS - 1) / BS, tilen = (colA + BS - 1) / BS;
    // the row pointer may be a slice of a larger matrix's (a row block: pointers not rebased): the block's nonzeros are [base, base + nnz) of the column / value arrays
    long long base = 0, nnz = 0;
    if (csr_on_device) {   // (device CSR: based at 0 by contract; its last row pointer is the one number fetched)
        int last = 0;
        TC_TRY(hipMemcpy(&last, h_rowptr + rowA, sizeof(int), hipMemcpyDeviceToHost));
        nnz = last;
    } else { base = h_rowptr[0]; nnz = (long long)h_rowptr[rowA] - base; }
    std::vector<int> rebased;
    if (base != 0) {
        rebased.resize((size_t)rowA + 1);
        for (int r = 0; r <= rowA; r++) rebased[(size_t)r] = (int)(h_rowptr[r] - base);
        h_rowptr = rebased.data(); h_colidx += base; h_val += base;
    }
    D->rowA = rowA; D->colA = colA; D->nnz = nnz;
    T.tilem = tilem; T.tilen = tilen;
    const int cb_bits = bits_for(tilen), bi_bits = bits_for(tilem);
    D->cb_bits = cb_bits;

    // ---- the CSR arrays cross the bus (the only large upload of the device pipeline)
    std::vector<void *> later;
    struct Later {   // released on every way out, after the upload has been joined (members are destroyed in reverse order: `values` below goes first)
        std::vector<void *> &v;
        ~Later() { t_free_later = nullptr; for (void *q : v) (void)hipFree(q); v.clear(); }
    } later_guard{later};
    ValueUpload values;   // (joined on every way out of this function, before anybody frees d_val)
    double t0 = now_ms();
    int *d_rowptr = nullptr, *d_colidx = nullptr; val_t *d_val = nullptr;
    if (csr_on_device) { d_rowptr = const_cast<int *>(h_rowptr); d_colidx = const_cast<int *>(h_colidx); d_val = const_cast<val_t *>(h_val); }   // (borrowed: never in D->allocs)
    else {
        if (dalloc(D, &d_rowptr, (size_t)rowA + 1, false) || dalloc(D, &d_colidx, (size_t)nnz, false) || dalloc(D, &d_val, (size_t)nnz, false)) return -3;
        TC_TRY(hipMemcpy(d_rowptr, h_rowptr, ((size_t)rowA + 1) * sizeof(int), hipMemcpyHostToDevice));
        if (nnz) TC_TRY(hipMemcpy(d_colidx, h_colidx, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
        // the values follow on a stream of their own, fed by a helper thread (a copy from pageable memory holds its caller until the last chunk is staged), while this thread sorts the
        // keys, lists the tiles and selects their formats: nothing before the packing reads a value (config 4: 0.67 GB = 12.6 ms of bus time behind 7 ms of kernels)
        if (nnz) {
            int dev = 0;
            TC_TRY(hipGetDevice(&dev));
            bool started = false;
            try { values.start(dev, d_val, h_val, (size_t)nnz * sizeof(val_t)); started = true; } catch (...) {}   // (no thread to be had: the plain copy)
            if (started) t_free_later = &later;
            else TC_TRY(hipMemcpy(d_val, h_val, (size_t)nnz * sizeof(val_t), hipMemcpyHostToDevice));
        }
    }
    D->rowptr = d_rowptr; D->colidx = d_colidx; D->val = d_val;
    D->ms_upload = now_ms() - t0; t0 = now_ms();

    // ---- keys + stable sort by (tile-row, column block)
    u64 *key_a = nullptr, *key_b = nullptr; int *ent_a = nullptr, *ent_b = nullptr;
    if (dalloc(D, &key_a, (size_t)nnz, false) || dalloc(D, &key_b, (size_t)nnz, false) || dalloc(D, &ent_a, (size_t)nnz, false) || dalloc(D, &ent_b, (size_t)nnz, false)) return -3;
    hipLaunchKernelGGL(k_tc_keys, dim3(blocks_for(tilem, 4)), dim3(256), 0, 0, rowA, tilem, d_rowptr, d_colidx, 8 + cb_bits, key_a, ent_a);
    TC_TRY(hipGetLastError());
    {
        u64 *k_cur = key_a, *k_alt = key_b; int *v_cur = ent_a, *v_alt = ent_b;
        size_t tmp_b = 0; void *tmp = nullptr;
        if (nnz > 0) {
            TC_TRY(prims::sort_pairs_u64_int(nullptr, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)nnz, 8u, (unsigned)(8 + cb_bits + bi_bits), (hipStream_t)0));
            TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
            hipError_t e = prims::sort_pairs_u64_int(tmp, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)nnz, 8u, (unsigned)(8 + cb_bits + bi_bits), (hipStream_t)0);
            if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)0)   /* (not the device: the values' upload is still under way on its own stream) */;
            free_now_or_later(tmp);
            TC_TRY(e);
        }
        D->key = k_cur; D->ent = v_cur;
        dfree(D, k_alt); dfree(D, v_alt);
    }
    D->ms_sort = now_ms() - t0; t0 = now_ms();

    // ---- tile list
    int tilenum = 0;
    int *d_tile_ptr = nullptr;
    if (dalloc(D, &d_tile_ptr, (size_t)tilem + 1, true)) return -3;
    T.tile_ptr = d_tile_ptr;
    u64 *d_uniq = nullptr; int *d_counts = nullptr, *d_nruns = nullptr;
    if (nnz > 0) {
        if (dalloc(D, &d_uniq, (size_t)nnz, false) || dalloc(D, &d_counts, (size_t)nnz + 1, false) || dalloc(D, &d_nruns, 1, true)) return -3;
        size_t tmp_b = 0; void *tmp = nullptr;
        TC_TRY(prims::rle_u64(nullptr, tmp_b, D->key, KEY_TILE_SHIFT, (unsigned)nnz, d_uniq, d_counts, d_nruns, (hipStream_t)0));
        TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = prims::rle_u64(tmp, tmp_b, D->key, KEY_TILE_SHIFT, (unsigned)nnz, d_uniq, d_counts, d_nruns, (hipStream_t)0);
        if (e == hipSuccess) e = hipMemcpy(&tilenum, d_nruns, sizeof(int), hipMemcpyDeviceToHost);
        free_now_or_later(tmp);
        TC_TRY(e);
    }
    T.tilenum = tilenum;
    if (!(flags & TILESPMV_CREATE_QUIET)) printf("\n  The number of tile = %i\n", tilenum);
    const size_t np1 = (size_t)tilenum + 1;
    // ---- pool of the per-tile arrays: 3 + 9 + 7 arrays of tilenum (+ 1) elements
    if (pool_begin(D, 3 * pool_need(np1, 4) + 3 * pool_need(np1, 1) + 6 * pool_need(np1, 4) + 7 * pool_need(np1, 4) + pool_need(16, 8))) return -3;
    int *d_tile_columnidx = nullptr, *d_tile_nnz = nullptr, *d_tile_bi = nullptr;
    if (dalloc(D, &d_tile_columnidx, (size_t)tilenum, true) || dalloc(D, &d_tile_nnz, np1, true) || dalloc(D, &d_tile_bi, (size_t)tilenum, true)) return -3;
    T.tile_columnidx = d_tile_columnidx; T.tile_nnz = d_tile_nnz; D->tile_bi = d_tile_bi;
    if (tilenum > 0) {
        hipLaunchKernelGGL(k_tc_tiles, dim3(blocks_for(tilenum, 256)), dim3(256), 0, 0, tilenum, tilem, cb_bits, d_uniq, d_tile_columnidx, d_tile_bi, d_tile_ptr);
        TC_TRY(hipGetLastError());
        // tile_nnz = exclusive scan of the run lengths (np1 elements: the last one is the total)
        TC_TRY(hipMemsetAsync(d_counts + tilenum, 0, sizeof(int), 0));
        size_t tmp_b = 0; void *tmp = nullptr;
        TC_TRY(prims::scan_int(nullptr, tmp_b, d_counts, d_tile_nnz, np1, (hipStream_t)0));
        TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = prims::scan_int(tmp, tmp_b, d_counts, d_tile_nnz, np1, (hipStream_t)0);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)0);
        free_now_or_later(tmp);
        TC_TRY(e);
    }
    dfree(D, d_uniq); dfree(D, d_counts); dfree(D, d_nruns);
    D->ms_tiles = now_ms() - t0; t0 = now_ms();

    // ---- per-tile metadata + format selection + scans
    TileArrays A{};
    int *off7[7];
    if (dalloc(D, &A.Format, (size_t)tilenum, true) || dalloc(D, &A.blknnz, np1, true) || dalloc(D, &A.blknnznnz, np1, true) || dalloc(D, &A.dnsrowptr, np1, true) ||
        dalloc(D, &A.dnscolptr, np1, true) || dalloc(D, &A.tilewidth, (size_t)tilenum, true) || dalloc(D, &A.csrptr_offset, np1, true) || dalloc(D, &A.hyb_coocount, np1, true) ||
        dalloc(D, &A.new_coocount, np1, true) || dalloc(D, &A.hyb_bytes, np1, true))
        return -3;
    for (int f = 0; f < 7; f++) { if (dalloc(D, &off7[f], np1, true)) return -3; A.fmt_offset[f] = off7[f]; }
    T.Format = A.Format; T.blknnz = A.blknnz; T.blknnznnz = A.blknnznnz; T.dnsrowptr = A.dnsrowptr; T.dnscolptr = A.dnscolptr; T.tilewidth = A.tilewidth;
    T.csrptr_offset = A.csrptr_offset; T.hyb_coocount = A.hyb_coocount; T.new_coocount = A.new_coocount;
    T.csr_offset = off7[TILESPMV_FMT_CSR]; T.coo_offset = off7[TILESPMV_FMT_COO]; T.ell_offset = off7[TILESPMV_FMT_ELL]; T.hyb_offset = off7[TILESPMV_FMT_HYB];
    T.dns_offset = off7[TILESPMV_FMT_DNS]; T.dnsrow_offset = off7[TILESPMV_FMT_DNSROW]; T.dnscol_offset = off7[TILESPMV_FMT_DNSCOL];
    if (tilenum > 0) {
        hipLaunchKernelGGL(k_tc_select, dim3(blocks_for(tilenum, 256)), dim3(256), 0, 0, tilenum, tilem, tilen, rowA, colA, allow_hyb, cdna4, d_tile_nnz, d_tile_bi, d_tile_columnidx, D->key, A);
        TC_TRY(hipGetLastError());
    }
    ScanSet S{};
    int *scans[] = {T.csr_offset, T.csrptr_offset, T.coo_offset, T.ell_offset, T.hyb_offset, T.dns_offset, T.dnsrow_offset, T.dnscol_offset, T.dnsrowptr, T.dnscolptr, T.hyb_coocount, T.new_coocount, T.blknnz, A.hyb_bytes};
    static const char *names[] = {"csr_offset", "csrptr_offset", "coo_offset", "ell_offset", "hyb_offset", "dns_offset", "dnsrow_offset", "dnscol_offset", "dnsrowptr", "dnscolptr", "hyb_coocount", "new_coocount", "blknnz", "hybIdx bytes"};
    constexpr int NS = 14;
    for (int k = 0; k < NS; k++) S.a[k] = scans[k];
    unsigned long long *d_totals = nullptr, h_totals[NS] = {0};
    if (dalloc(D, &d_totals, NS, true)) return -3;
    hipLaunchKernelGGL(k_tc_totals, dim3((unsigned)std::min<long long>(2048, blocks_for((long long)np1, 256))), dim3(256), 0, 0, S, NS, (long long)np1, d_totals);
    TC_TRY(hipGetLastError());
    TC_TRY(hipMemcpy(h_totals, d_totals, sizeof(h_totals), hipMemcpyDeviceToHost));
    dfree(D, d_totals);
    for (int k = 0; k < NS; k++)
        if (h_totals[k] > (unsigned long long)INT32_MAX) { fprintf(stderr, "tilespmv: %s exceeds the int32 offsets of Tile_matrix\n", names[k]); return -2; }
    {
        size_t tmp_b = 0; void *tmp = nullptr;
        TC_TRY(prims::scan_int(nullptr, tmp_b, scans[0], scans[0], np1, (hipStream_t)0));
        TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = hipSuccess;
        for (int k = 0; k < NS && e == hipSuccess; k++)
            if (h_totals[k] > 0) e = prims::scan_int(tmp, tmp_b, scans[k], scans[k], np1, (hipStream_t)0);   // (an all-zero array is its own scan)
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)0);
        free_now_or_later(tmp);
        TC_TRY(e);
    }
    T.csrsize = (int)h_totals[0]; T.csrptrlen = (int)h_totals[1]; T.coosize = (int)h_totals[2]; T.ellsize = (int)h_totals[3];
    T.hybsize = (int)h_totals[4]; T.hybcoosize = (int)h_totals[10]; T.hybellsize = T.hybsize - T.hybcoosize;   // (a HYB tile stores its ELL part + its remainder)
    D->hyb_byte_off = A.hyb_bytes;   // (scanned: per tile its first byte in hybIdx)
    if (T.hybsize > 0) {             // ... and as the 64-bit offsets the plan builders' per-tile functions take (plan_tile_ops.h hyb_off)
        long long *d_off64 = nullptr;
        if (dalloc(D, &d_off64, np1, true)) return -3;
        hipLaunchKernelGGL(k_tc_widen, dim3(blocks_for((long long)np1, 256)), dim3(256), 0, 0, (long long)np1, (const int *)A.hyb_bytes, d_off64);
        TC_TRY(hipGetLastError());
        D->hyb_off = d_off64;
    }
    T.dnssize = (int)h_totals[5]; T.dnsrowsize = (int)h_totals[6]; T.dnscolsize = (int)h_totals[7]; T.coototal = (int)h_totals[11];
    const int ndenserow = (int)h_totals[8], ndensecol = (int)h_totals[9];
    D->ms_select = now_ms() - t0; t0 = now_ms();
    pool_end();

    // ---- payload arrays: one pool
    {
        const size_t sv = sizeof(val_t);
        size_t bytes = pool_need((size_t)T.csrsize, sv) + pool_need((size_t)T.csrptrlen, 1) + pool_need((size_t)T.csrsize, 1) + pool_need(((size_t)T.csrsize + 1) / 2, 1) + pool_need((size_t)T.coosize, sv) +
                       pool_need((size_t)T.coosize, 1) + pool_need((size_t)T.ellsize, sv) + pool_need((size_t)T.ellsize, 1) + pool_need(((size_t)T.ellsize + 1) / 2, 1) + pool_need((size_t)T.hybsize + 1, sv) + pool_need(((size_t)T.hybellsize + 1) / 2 + (size_t)T.hybcoosize + (size_t)tilem + 8, 1) +
                       pool_need((size_t)T.dnssize, sv) + pool_need((size_t)T.dnsrowsize, sv) + pool_need((size_t)ndenserow, 1) + pool_need((size_t)T.dnscolsize, sv) + pool_need((size_t)ndensecol, 1);
        if (want_deferred) bytes += pool_need((size_t)rowA + 1, 4) + pool_need((size_t)T.coototal, 4) + pool_need((size_t)T.coototal, sv) + 2 * pool_need((size_t)T.coototal, 4) + pool_need((size_t)T.coototal, sv);
        if (pool_begin(D, bytes)) return -3;
    }
    PackArrays P{};
    unsigned char *d_csr_idx = nullptr, *d_ell_idx = nullptr, *d_hybidx = nullptr; val_t *d_hybval = nullptr;
    if (dalloc(D, &P.Blockcsr_Val, (size_t)T.csrsize, true) || dalloc(D, &P.Blockcsr_Ptr, (size_t)T.csrptrlen, true) || dalloc(D, &P.csr_col, (size_t)T.csrsize, true) ||
        dalloc(D, &d_csr_idx, ((size_t)T.csrsize + 1) / 2, true) || dalloc(D, &P.Blockcoo_Val, (size_t)T.coosize, true) || dalloc(D, &P.coo_compressed_Idx, (size_t)T.coosize, true) ||
        dalloc(D, &P.Blockell_Val, (size_t)T.ellsize, true) || dalloc(D, &P.ell_col, (size_t)T.ellsize, true) || dalloc(D, &d_ell_idx, ((size_t)T.ellsize + 1) / 2, true) ||
        dalloc(D, &d_hybval, (size_t)T.hybsize + 1, true) || dalloc(D, &d_hybidx, ((size_t)T.hybellsize + 1) / 2 + (size_t)T.hybcoosize + (size_t)tilem + 8, true) || dalloc(D, &P.Blockdense_Val, (size_t)T.dnssize, true) ||
        dalloc(D, &P.Blockdenserow_Val, (size_t)T.dnsrowsize, true) || dalloc(D, &P.denserowid, (size_t)ndenserow, true) || dalloc(D, &P.Blockdensecol_Val, (size_t)T.dnscolsize, true) ||
        dalloc(D, &P.densecolid, (size_t)ndensecol, true))
        return -3;
    T.Blockcsr_Val = P.Blockcsr_Val; T.Blockcsr_Ptr = P.Blockcsr_Ptr; T.csr_compressedIdx = d_csr_idx;
    T.Blockcoo_Val = P.Blockcoo_Val; T.coo_compressed_Idx = P.coo_compressed_Idx;
    T.Blockell_Val = P.Blockell_Val; T.ell_compressedIdx = d_ell_idx;
    T.Blockhyb_Val = d_hybval; T.hybIdx = d_hybidx;
    P.Blockhyb_Val = d_hybval; P.hybIdx = d_hybidx; P.hyb_byte_off = A.hyb_bytes;
    T.Blockdense_Val = P.Blockdense_Val; T.Blockdenserow_Val = P.Blockdenserow_Val; T.denserowid = P.denserowid;
    T.Blockdensecol_Val = P.Blockdensecol_Val; T.densecolid = P.densecolid;
    int *d_dptr = nullptr, *d_dcol = nullptr; val_t *d_dval = nullptr;
    if (want_deferred) {
        if (dalloc(D, &d_dptr, (size_t)rowA + 1, true) || dalloc(D, &d_dcol, (size_t)T.coototal, true) || dalloc(D, &d_dval, (size_t)T.coototal, true) ||
            dalloc(D, &P.x_key, (size_t)T.coototal, true) || dalloc(D, &P.x_col, (size_t)T.coototal, true) || dalloc(D, &P.x_val, (size_t)T.coototal, true))
            return -3;
        P.deferredcoo_ptr = d_dptr;
        T.deferredcoo_ptr = d_dptr; T.deferredcoo_colidx = d_dcol; T.deferredcoo_val = d_dval;
    }
    pool_end();
    TC_TRY(values.wait());   // the packing reads the values
    t_free_later = nullptr;
    for (void *q : later) (void)hipFree(q);
    later.clear();
    if (tilenum > 0) {
        hipLaunchKernelGGL(k_tc_pack, dim3(blocks_for(tilenum, 256)), dim3(256), 0, 0, tilenum, tilem, tilen, rowA, colA, T, d_tile_bi, D->key, D->ent, d_colidx, d_val, P);
        TC_TRY(hipGetLastError());
    }
    if (T.csrsize > 0) hipLaunchKernelGGL(k_tc_nibbles, dim3(blocks_for(((long long)T.csrsize + 1) / 2, 256)), dim3(256), 0, 0, P.csr_col, d_csr_idx, (long long)T.csrsize);
    if (T.ellsize > 0) hipLaunchKernelGGL(k_tc_nibbles, dim3(blocks_for(((long long)T.ellsize + 1) / 2, 256)), dim3(256), 0, 0, P.ell_col, d_ell_idx, (long long)T.ellsize);
    TC_TRY(hipGetLastError());
    if (want_deferred) {
        // rows of the extracted matrix: counts -> pointers; entries: stable sort of the tile-ordered list by row = "order of appearance" inside every row (src/csr2tile.h:943-950)
        size_t tmp_b = 0; void *tmp = nullptr;
        TC_TRY(prims::scan_int(nullptr, tmp_b, d_dptr, d_dptr, (size_t)rowA + 1, (hipStream_t)0));
        TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = prims::scan_int(tmp, tmp_b, d_dptr, d_dptr, (size_t)rowA + 1, (hipStream_t)0);
        (void)hipDeviceSynchronize();
        (void)hipFree(tmp);
        TC_TRY(e);
        if (T.coototal > 0) {
            unsigned *key_b2 = nullptr; int *pos_a = nullptr, *pos_b = nullptr;
            if (dalloc(D, &key_b2, (size_t)T.coototal, false) || dalloc(D, &pos_a, (size_t)T.coototal, false) || dalloc(D, &pos_b, (size_t)T.coototal, false)) return -3;
            hipLaunchKernelGGL(k_tc_iota, dim3(blocks_for(T.coototal, 256)), dim3(256), 0, 0, T.coototal, pos_a);
            TC_TRY(hipGetLastError());
            unsigned *k_cur = P.x_key, *k_alt = key_b2; int *v_cur = pos_a, *v_alt = pos_b;
            const unsigned row_bits = (unsigned)bits_for(std::max(rowA, 2));
            tmp_b = 0; tmp = nullptr;
            TC_TRY(prims::sort_pairs_u32_int(nullptr, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)T.coototal, 0u, row_bits, (hipStream_t)0));
            TC_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
            e = prims::sort_pairs_u32_int(tmp, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)T.coototal, 0u, row_bits, (hipStream_t)0);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_tc_deferred, dim3(blocks_for(T.coototal, 256)), dim3(256), 0, 0, T.coototal, (const int *)v_cur, (const int *)P.x_col, (const val_t *)P.x_val, d_dcol, d_dval);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(tmp);
            TC_TRY(e);
            dfree(D, key_b2); dfree(D, pos_a); dfree(D, pos_b);
            int *d_cnt = nullptr;
            if (dalloc(D, &d_cnt, 1, true)) return -3;
            hipLaunchKernelGGL(k_tc_unsorted_rows, dim3(blocks_for(rowA, 256)), dim3(256), 0, 0, rowA, (const int *)d_dptr, (const int *)d_dcol, d_cnt);
            TC_TRY(hipGetLastError());
            TC_TRY(hipMemcpy(&D->unsorted_rows, d_cnt, sizeof(int), hipMemcpyDeviceToHost));
            dfree(D, d_cnt);
        }
        dfree(D, P.x_key); dfree(D, P.x_col); dfree(D, P.x_val);
        D->have_deferred = true;
    }
    TC_TRY(hipDeviceSynchronize());
    dfree(D, P.csr_col); dfree(D, P.ell_col);
    D->ms_pack = now_ms() - t0;
    if (verbose)
        fprintf(stderr, "tilespmv: device Tile_create: upload of the index arrays %.1f ms (the values cross the bus behind the next three steps), keys + sort %.1f, tile list %.1f, selection + scans %.1f, packing %.1f (%d tiles, %lld nonzeros)\n", D->ms_upload, D->ms_sort, D->ms_tiles,
                D->ms_select, D->ms_pack, tilenum, nnz);
    return 0;
}

}  // namespace

int devtile_create(DevTile **out, int rowA, int colA, const MAT_PTR_TYPE *h_rowptr, const int *h_colidx, const val_t *h_val, unsigned flags, bool want_deferred, bool csr_on_device)
{
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); fprintf(stderr, "tilespmv: no HIP device visible — the device Tile_create has no CPU fallback (use Tile_create)\n"); return -1; }
    DevTile *D = new DevTile();
    const int rc = create_impl(D, rowA, colA, h_rowptr, h_colidx, h_val, flags, want_deferred, csr_on_device);
    pool_end();
    if (rc != 0) { devtile_destroy(D); return rc; }
    *out = D;
    return 0;
}

void devtile_destroy(DevTile *D)
{
    if (!D) return;
    for (void *p : D->allocs) (void)hipFree(p);
    for (void *p : D->pools) (void)hipFree(p);
    delete D;
}



int devtile_download(const DevTile *D, Tile_matrix *H)
{
    const Tile_matrix &T = D->T;
    memset(H, 0, sizeof(*H));
    H->tilem = T.tilem; H->tilen = T.tilen; H->tilenum = T.tilenum;
    H->csrsize = T.csrsize; H->csrptrlen = T.csrptrlen; H->coosize = T.coosize; H->ellsize = T.ellsize; H->hybsize = T.hybsize; H->hybellsize = T.hybellsize; H->hybcoosize = T.hybcoosize;
    H->dnssize = T.dnssize; H->dnsrowsize = T.dnsrowsize; H->dnscolsize = T.dnscolsize; H->coototal = T.coototal;
    const size_t tn = (size_t)T.tilenum, np1 = tn + 1;
    int rc = 0;
    auto get = [&](auto **dst, const auto *src, size_t n) {
        typedef typename std::remove_pointer<typename std::remove_reference<decltype(*dst)>::type>::type V;
        *dst = zalloc<V>(n);
        if (n && src && hipMemcpy(*dst, src, n * sizeof(V), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); rc = -3; }
    };
    get(&H->tile_ptr, T.tile_ptr, (size_t)T.tilem + 1);
    get(&H->tile_columnidx, T.tile_columnidx, tn);
    get(&H->tile_nnz, T.tile_nnz, np1);
    get(&H->Format, T.Format, tn);
    get(&H->blknnz, T.blknnz, np1);
    get(&H->blknnznnz, T.blknnznnz, np1);
    get(&H->dnsrowptr, T.dnsrowptr, np1);
    get(&H->dnscolptr, T.dnscolptr, np1);
    get(&H->tilewidth, T.tilewidth, tn);
    get(&H->csr_offset, T.csr_offset, np1);
    get(&H->csrptr_offset, T.csrptr_offset, np1);
    get(&H->coo_offset, T.coo_offset, np1);
    get(&H->ell_offset, T.ell_offset, np1);
    get(&H->hyb_offset, T.hyb_offset, np1);
    get(&H->hyb_coocount, T.hyb_coocount, np1);
    get(&H->dns_offset, T.dns_offset, np1);
    get(&H->dnsrow_offset, T.dnsrow_offset, np1);
    get(&H->dnscol_offset, T.dnscol_offset, np1);
    get(&H->new_coocount, T.new_coocount, np1);
    get(&H->Blockcsr_Val, T.Blockcsr_Val, (size_t)T.csrsize);
    get(&H->Blockcsr_Ptr, T.Blockcsr_Ptr, (size_t)T.csrptrlen);
    get(&H->csr_compressedIdx, T.csr_compressedIdx, ((size_t)T.csrsize + 1) / 2);
    get(&H->Blockcoo_Val, T.Blockcoo_Val, (size_t)T.coosize);
    get(&H->coo_compressed_Idx, T.coo_compressed_Idx, (size_t)T.coosize);
    get(&H->Blockell_Val, T.Blockell_Val, (size_t)T.ellsize);
    get(&H->ell_compressedIdx, T.ell_compressedIdx, ((size_t)T.ellsize + 1) / 2);
    get(&H->Blockhyb_Val, T.Blockhyb_Val, (size_t)T.hybellsize + (size_t)T.hybcoosize);
    get(&H->hybIdx, T.hybIdx, ((size_t)T.hybellsize + 1) / 2 + (size_t)T.hybcoosize + (size_t)T.tilem + 8);
    get(&H->Blockdense_Val, T.Blockdense_Val, (size_t)T.dnssize);
    get(&H->Blockdenserow_Val, T.Blockdenserow_Val, (size_t)T.dnsrowsize);
    int ndr = 0, ndc = 0;
    if (tn) { ndr = H->dnsrowptr[tn]; ndc = H->dnscolptr[tn]; }
    get(&H->denserowid, T.denserowid, (size_t)ndr);
    get(&H->Blockdensecol_Val, T.Blockdensecol_Val, (size_t)T.dnscolsize);
    get(&H->densecolid, T.densecolid, (size_t)ndc);
    if (D->have_deferred) {
        get(&H->deferredcoo_val, T.deferredcoo_val, (size_t)T.coototal);
        get(&H->deferredcoo_colidx, T.deferredcoo_colidx, (size_t)T.coototal);
        get(&H->deferredcoo_ptr, T.deferredcoo_ptr, (size_t)D->rowA + 1);
        if (rc == 0 && D->unsorted_rows > 0)   // rows whose columns do not increase (unsorted CSR input): the reference's pivot sort, on the host (src/csr2tile.h:951-958, src/utils.h:103-137)
            parallel_chunks(D->rowA, 4096, [&](int64_t b, int64_t e, int) {
                for (int64_t r = b; r < e; r++) {
                    const int p = H->deferredcoo_ptr[r], len = H->deferredcoo_ptr[r + 1] - p;
                    int *k = H->deferredcoo_colidx + p;
                    bool increasing = true;
                    for (int i = 1; i < len && increasing; i++) increasing = k[i - 1] < k[i];
                    if (!increasing) pivot_sort(k, H->deferredcoo_val + p, len);
                }
            });
    } else {
        H->deferredcoo_val = zalloc<val_t>(0); H->deferredcoo_colidx = zalloc<int>(0); H->deferredcoo_ptr = zalloc<int>((size_t)D->rowA + 1);
    }
    return rc;
}

}  // namespace tilespmv

extern "C" int Tile_create_device(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, const MAT_PTR_TYPE *csrRowPtrA, const int *csrColIdxA, const MAT_VAL_TYPE *csrValA, unsigned flags)
{
    (void)nnzA;   // like Tile_create, the row pointer decides how many nonzeros are used
    tilespmv::DevTile *D = nullptr;
    int rc = tilespmv::devtile_create(&D, rowA, colA, csrRowPtrA, csrColIdxA, csrValA, flags, true);
    if (rc != 0) return rc;
    rc = tilespmv::devtile_download(D, matrix);
    tilespmv::devtile_destroy(D);
    if (rc != 0) Tile_destroy(matrix);
    return rc;
}
