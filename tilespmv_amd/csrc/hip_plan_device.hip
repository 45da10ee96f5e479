// hip_plan_device.hip — COUNT / EMIT / ENCODE of the unit-stream builder as kernels over a device-resident Tile_matrix (hip_plan_device.h).
// One thread per tile (one per tile-row for the pooled windows, which run across a tile-row's tiles) calling the functions of plan_tile_ops.h that the host builder
// calls from its loops: the same records at the same offsets.  The host keeps the decisions (hip_plan_stream.hip CHOOSE / CUT / ORDER) and sends back prefix arrays.
#include <cstring>

#include <sys/time.h>

#include <hip/hip_runtime.h>
#include "hip_prims.h"

#include "hip_plan_device.h"

namespace tilespmv {
namespace {

#define PD_TRY(expr)                                                                                                \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) {                                                                                     \
            fprintf(stderr, "tilespmv: device plan build: HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                                                \
            return -3;                                                                                              \
        }                                                                                                           \
    } while (0)

typedef unsigned long long u64;
inline unsigned nblk(long long n, int per) { return (unsigned)std::max<long long>(1, (n + per - 1) / per); }

// a temporary device array that frees itself
template <class V>
struct Tmp {
    V *p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n, bool zero)
    {
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(V) + 16);
        if (e == hipSuccess && zero) e = hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(V) + 16, 0);
        return e;
    }
    hipError_t from(const std::vector<V> &h) { hipError_t e = alloc(h.size(), false); if (e == hipSuccess && !h.empty()) e = hipMemcpy(p, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice); return e; }
};

hipError_t scan_in_place(int *a, size_t n)
{
    size_t tmp_b = 0; void *tmp = nullptr;
    hipError_t e = prims::scan_int(nullptr, tmp_b, a, a, n, (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = hipMalloc(&tmp, std::max<size_t>(tmp_b, 16));
    if (e != hipSuccess) return e;
    e = prims::scan_int(tmp, tmp_b, a, a, n, (hipStream_t)0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(tmp);
    return e;
}

__global__ __launch_bounds__(256) void k_pd_gather_ints(const int *__restrict__ a, const long long *__restrict__ idx, int n, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[idx[i]];
}

// ---- COUNT
__global__ __launch_bounds__(256) void k_pd_count_tiles(const Tile_matrix T, const int *__restrict__ tile_bi, const long long *__restrict__ hyb_off, int t_begin, int nt, int rowA, int colA, bool coo_in_tile, bool dense_mfma, int csr_form, bool absorb,
                                                          int *__restrict__ tu, int *__restrict__ tc, int *__restrict__ td, int *__restrict__ tp)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nt) return;
    const int t = t_begin + (int)gid, bi = tile_bi[t], rowlen = tile_rowlen(bi, T.tilem, rowA);
    const TileCount k = tile_count(&T, t, rowlen, T.tilen, colA, coo_in_tile, dense_mfma, csr_form, absorb, T.tile_ptr[bi], T.tile_ptr[bi + 1]);
    tu[gid] = k.nunits; tc[gid] = k.ncoo; td[gid] = k.ndense;
    if (csr_form >= 2) tp[gid] = pool_one_tile_count(&T, t, rowlen, coo_in_tile, hyb_off);   // what the tile adds to its tile-row's pool
}
// The pooled part (round 5, second half: one thread per tile-row took 0.35 s per pass on R-MAT 21 x 16, whose hub tile-rows hold 10^5 nonzeros in 10^4-10^5 tiny tiles each — and a
// wavefront waits for its heaviest lane).  Two balanced steps instead: (1) every tile writes its contribution where the exclusive scan of the contributions (tp) puts it — the tiles of a
// tile-row back to back, which is the array the host builder's pool_row makes — one thread per TILE with the host builder's per-tile function; (2) one WAVEFRONT per tile-row walks
// the windows (pool_windows_wave below).
__global__ __launch_bounds__(256) void k_pd_fill_pool(const Tile_matrix T, const int *__restrict__ tile_bi, const long long *__restrict__ hyb_off, int t_begin, int nt, int rowA, bool coo_in_tile, const int *__restrict__ tp, PoolEnt *__restrict__ pool)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nt) return;
    if (tp[gid + 1] == tp[gid]) return;
    const int t = t_begin + (int)gid;
    (void)pool_one_tile(&T, t, tile_rowlen(tile_bi[t], T.tilem, rowA), coo_in_tile, hyb_off, pool + tp[gid]);
}
// The window walk (plan_tile_ops.h pool_windows) by a wavefront, 64 nonzeros per step: every lane finds where a window starting at ITS nonzero would end (at most 15 look-aheads),
// then the wavefront follows the chain of window starts through the step's 64 candidates with register reads (v_readlane: a few cycles per window, against a dependent
// global load per nonzero when one lane walks alone — 80 ms per pass for the hub tile-rows of R-MAT 21 x 16).  f(is_start, begin, end, lane) runs on all 64 lanes
// with the step's windows marked; windows are visited in pool order across steps, so prefix counts over (steps, lanes) number them exactly as the sequential walk does.
template <class F>
__device__ __forceinline__ void pool_windows_wave(const PoolEnt *s, long long n, unsigned width, int lane, F f)
{
    long long cur = 0;   // next window start (wavefront-uniform)
    for (long long c0 = 0; c0 < n; c0 += 64) {
        const long long i = c0 + lane;
        int len = 0;
        if (i < n) {   // first of the (at most 15) following nonzeros whose column is `width` or more above this one's: columns ascend, so a binary search finds it
            const unsigned long long lim = (unsigned long long)s[i].col + width;
            long long lo = i + 1, hi = i + 16 < n ? i + 16 : n;   // answer in [lo, hi]
            while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((unsigned long long)s[mid].col < lim) lo = mid + 1; else hi = mid; }
            len = (int)(lo - i);
        }
        unsigned long long starts = 0;
        int at = __builtin_amdgcn_readfirstlane((int)(cur - c0));   // 0 .. 15: the last window of the step before may reach into this one
        while (at < 64 && c0 + at < n) {
            starts |= 1ull << at;
            at += __builtin_amdgcn_readlane(len, at);
        }
        cur = c0 + at;
        f((starts >> lane) & 1ull, i, i + len, lane);
    }
}
__device__ __forceinline__ int wave_sum(int v)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__global__ __launch_bounds__(256) void k_pd_count_pool(const int *__restrict__ tile_ptr, int tr0, int ntr, int t_begin, unsigned width, const int *__restrict__ tp, const PoolEnt *__restrict__ pool, int *__restrict__ pool_u, int *__restrict__ pool_c, unsigned long long *__restrict__ stat /* [0] units, [1] lines */)
{
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= ntr) return;   // (whole wavefronts leave together)
    const int a = tile_ptr[tr0 + i] - t_begin, b = tile_ptr[tr0 + i + 1] - t_begin;
    const PoolEnt *scratch = pool + tp[a];
    const long long n = tp[b] - tp[a];
    int nu = 0, nc = 0, nl = 0;
    pool_windows_wave(scratch, n, width, lane, [&](bool start, long long b, long long e, int) {
        if (!start) return;
        if (e - b >= POOL_MIN_FILL) { nu++; nl += pool_window_lines(scratch, b, e); }
        else nc += (int)(e - b);
    });
    nu = wave_sum(nu); nc = wave_sum(nc); nl = wave_sum(nl);
    if (lane != 0) return;
    pool_u[i] = nu; pool_c[i] = nc;
    if (nu) { atomicAdd(&stat[0], (unsigned long long)nu); atomicAdd(&stat[1], (unsigned long long)nl); }
}
__global__ __launch_bounds__(256) void k_pd_row_counts(const int *__restrict__ tile_ptr, int tr0, int ntr, int t_begin, const int *__restrict__ tu, const int *__restrict__ tc, const int *__restrict__ td,
                                                         const int *__restrict__ pool_u, const int *__restrict__ pool_c, int *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntr) return;
    const int a = tile_ptr[tr0 + i] - t_begin, b = tile_ptr[tr0 + i + 1] - t_begin;
    out[3 * i] = tu[b] - tu[a] + (pool_u ? pool_u[i] : 0);
    out[3 * i + 1] = tc[b] - tc[a] + (pool_c ? pool_c[i] : 0);
    out[3 * i + 2] = td[b] - td[a];
}

// the sample of hip_plan_stream.hip StreamBuilder::count(), literally (tiles from last_first_tile on belong to the last tile-row, of last_rowlen rows)
__global__ __launch_bounds__(256) void k_pd_pattern_sample(const Tile_matrix T, int t_begin, int t_end, int step, int nsample, int last_first_tile, int last_rowlen, u64 *__restrict__ pats, int *__restrict__ npat)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsample) return;
    const int t = t_begin + s * step;
    int n = 0;
    if (t < t_end) {
        const int fmt = T.Format[t], rl = t >= last_first_tile ? last_rowlen : 16;
        if (fmt == TILESPMV_FMT_ELL) {
            const int off = T.ell_offset[t], w = T.tilewidth[t];
            for (int sl = 0; sl < w; sl++) { u64 nibs = 0; for (int r = 0; r < rl; r++) nibs |= (u64)nib_at(T.ell_compressedIdx, (long long)off + sl * rl + r) << (60 - 4 * r); pats[(size_t)s * 16 + n++] = nibs; }
        } else if (fmt == TILESPMV_FMT_CSR) {
            const int off = T.csr_offset[t], stored = T.blknnz[t + 1] - T.blknnz[t];
            const unsigned char *ptr = T.Blockcsr_Ptr + T.csrptr_offset[t];
            u64 nibs = 0;
            for (int r = 0; r < rl; r++) { const int k0 = ptr[r], k1 = r == rl - 1 ? stored : ptr[r + 1]; if (k1 > k0) nibs |= (u64)nib_at(T.csr_compressedIdx, (long long)off + k0) << (60 - 4 * r); }
            pats[(size_t)s * 16 + n++] = nibs;
        }
    }
    npat[s] = n;
}

// ---- EMIT
__global__ __launch_bounds__(256) void k_pd_fill_urow(uint2 *__restrict__ urow, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) urow[i] = make_uint2(0x01234567u, 0x89ABCDEFu);   // (units that keep one row per lane: identity)
}
__global__ __launch_bounds__(256) void k_pd_emit_tiles(const Tile_matrix T, const int *__restrict__ tile_bi, const long long *__restrict__ hyb_off, int t_begin, int nt, int tr0, int rowA, int colA, bool coo_in_tile, bool dense_mfma, int csr_form, bool absorb, bool derive,
                                                         const int *__restrict__ tu, const int *__restrict__ tc, const int *__restrict__ td, const int *__restrict__ pu, const int *__restrict__ pc,
                                                         const int *__restrict__ pd, const unsigned char *__restrict__ row_k, const unsigned char *__restrict__ row_split, const EmitOut O)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= nt) return;
    const int t = t_begin + (int)gid, bi = tile_bi[t], i = bi - tr0, a = T.tile_ptr[bi] - t_begin;
    EmitPos p{(long long)pu[i] + tu[gid] - tu[a], (long long)pc[i] + tc[gid] - tc[a], (long long)pd[i] + td[gid] - td[a]};
    const long long u0 = p.u;
    tile_emit(&T, t, tile_rowlen(bi, T.tilem, rowA), T.tilen, colA, coo_in_tile, dense_mfma, csr_form, (unsigned)row_k[i], hyb_off, O, p, absorb, T.tile_ptr[bi], T.tile_ptr[bi + 1], derive && !row_split[i]);
    // the tile that emits the last unit of an unsplit tile-row of a classic plan marks it (the kernel writes y there)
    if (csr_form < 2 && !row_split[i] && p.u > u0 && p.u == pu[i + 1]) { O.udesc[p.u - 1].x |= UNIT_EOR << UNIT_FLAG_SHIFT; O.udesc[p.u - 1].z |= UNIT_EOR << UNIT_FLAG_SHIFT; }
}
__global__ __launch_bounds__(256) void k_pd_emit_pool(const int *__restrict__ tile_ptr, int tr0, int ntr, int t_begin, unsigned width, const int *__restrict__ tp, const PoolEnt *__restrict__ pool, const int *__restrict__ tu,
                                                        const int *__restrict__ tc, const int *__restrict__ pu, const int *__restrict__ pc, const unsigned char *__restrict__ row_k, const EmitOut O)
{
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // one wavefront per tile-row, over the pool the counting pass filled
    const int lane = threadIdx.x & 63;
    if (i >= ntr) return;
    const int a = tile_ptr[tr0 + i] - t_begin, b = tile_ptr[tr0 + i + 1] - t_begin;
    const PoolEnt *scratch = pool + tp[a];
    const long long n = tp[b] - tp[a];
    long long u = (long long)pu[i] + tu[b] - tu[a], c = (long long)pc[i] + tc[b] - tc[a];   // behind what the tile-row's tiles emitted themselves
    const bool wide = width > 16u;
    const unsigned kr = (unsigned)row_k[i];
    pool_windows_wave(scratch, n, width, lane, [&](bool start, long long wb, long long we, int ln) {
        const bool unit = start && we - wb >= POOL_MIN_FILL;
        const unsigned long long units = __ballot(unit);
        const int ents = start && !unit ? (int)(we - wb) : 0;
        int incl = ents;
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d, 64); if (ln >= d) incl += up; }
        if (start) pool_window_put(scratch, wb, we, wide, kr, O, u + __popcll(units & ((1ull << ln) - 1ull)), c + (incl - ents));
        u += __popcll(units);
        c += __shfl(incl, 63, 64);
    });
}
__global__ __launch_bounds__(256) void k_pd_word0(const uint4 *__restrict__ udesc, long long n, unsigned *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = udesc[i].x;
}

// ---- ENCODE
__global__ __launch_bounds__(256) void k_pd_pack_desc(const uint4 *__restrict__ udesc, const uint2 *__restrict__ urow, const uint4 *__restrict__ ucol, const int4 *__restrict__ map, int ntasks, UDesc *__restrict__ packed,
                                                        URow *__restrict__ packed_row, uint4 *__restrict__ packed_col)
{
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const int4 m = map[t];
        for (int j = threadIdx.x; j < m.z; j += 256) {
            const uint4 d = udesc[(long long)m.x + j];
            packed[(long long)m.y + j] = UDesc{d.x, d.y, d.w};
            if (urow) { const uint2 r = urow[(long long)m.x + j]; packed_row[(long long)m.y + j] = URow{r.x, r.y}; }
            if (ucol) packed_col[(long long)m.y + j] = ucol[(long long)m.x + j];
        }
    }
}
// the nibble patterns of the units whose shift code (word 0 >> UNIT_SHIFT_SHIFT: units that took list entries, plan_tile_ops.h) is `code`, appended to out in any order (they are sorted next);
// all_same: every unit has that code (plans without such units: code 0) -> position i, no counter
__global__ __launch_bounds__(256) void k_pd_patterns(const UDesc *__restrict__ packed, long long n, unsigned code, bool all_same, u64 *__restrict__ out, unsigned long long *__restrict__ count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const UDesc d = packed[i];
    if (all_same) { out[i] = ((u64)d.n0 << 32) | d.n1; return; }
    if ((d.w0 >> UNIT_SHIFT_SHIFT) == code) out[atomicAdd(count, 1ull)] = ((u64)d.n0 << 32) | d.n1;
}
__global__ __launch_bounds__(256) void k_pd_shift_hist(const UDesc *__restrict__ packed, long long n, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned h[8];
    if (threadIdx.x < 8) h[threadIdx.x] = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) atomicAdd(&h[packed[i].w0 >> UNIT_SHIFT_SHIFT], 1u);
    __syncthreads();
    if (threadIdx.x < 8 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_pd_compact(const UDesc *__restrict__ packed, long long n, const uint4 *__restrict__ dict, DictRanges ranges, int cb_bits, unsigned *__restrict__ compact)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const UDesc d = packed[i];
    const u64 key = ((u64)d.n0 << 32) | d.n1;
    const unsigned code = d.w0 >> UNIT_SHIFT_SHIFT;
    int lo = ranges.off[code], hi = ranges.off[code + 1];   // lower bound among the entries of the unit's shift code (ascending nibbles)
    while (lo < hi) { const int mid = (lo + hi) >> 1; const uint4 q = dict[mid]; if ((((u64)q.x << 32) | q.y) < key) lo = mid + 1; else hi = mid; }
    compact[i] = (d.w0 & ((1u << cb_bits) - 1u)) | ((unsigned)lo << cb_bits) | ((d.w0 >> UNIT_FLAG_SHIFT) << 27);
}


// ---- ENTRIES
__global__ __launch_bounds__(256) void k_pe_keys(const STask *__restrict__ tasks, int ntasks, const int *__restrict__ ofs, int GS, int slab_shift, const int *__restrict__ ccol, const unsigned char *__restrict__ crow,
                                                  u64 *__restrict__ key, int *__restrict__ src, unsigned short *__restrict__ dest_q)
{
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const int b = tasks[t].coo_begin, e = tasks[t].coo_end, o = ofs[t];
        const u64 grp = (u64)(t / GS) << 32;
        const unsigned strip = (unsigned)(t & (GS - 1)) << slab_shift;
        for (int q = b + threadIdx.x; q < e; q += 256) {
            key[o + (q - b)] = grp | (unsigned)ccol[q];
            src[o + (q - b)] = q;
            dest_q[q] = (unsigned short)(strip | crow[q]);
        }
    }
}
// The chunk walk (plan_tile_ops.h pack_chunks) by a wavefront: one chunk of ECHUNK = 64 records per step, one record per lane.  Columns ascend along a list, so the entries that
// fit the chunk's column range are a prefix of the next 64: its length is the first zero of a ballot.  f(begin, count, base, padded) runs on all 64 lanes.
static_assert(ECHUNK == 64, "one record of a chunk per lane");
template <class F>
__device__ __forceinline__ void pack_chunks_wave(const u64 *__restrict__ K, long long n, int dest_bits, int lane, F f)
{
    const u64 span = 1ull << (32 - dest_bits);
    long long i = 0;
    while (i < n) {   // (wavefront-uniform)
        const unsigned b = (unsigned)K[i];
        const long long q = i + lane;
        const bool in = q < n && (u64)(unsigned)K[q] - b < span;
        const u64 out = ~__ballot(in);
        const int cnt = out ? __ffsll((unsigned long long)out) - 1 : 64;   // >= 1: the chunk's first entry defines the base
        f(i, cnt, b, i + cnt < n);   // interior chunks are filled up with null records
        i += cnt;
    }
}
// one wavefront per group: sizes of its packed list (+ the plan fact "entries far from the group's own rows")
__global__ __launch_bounds__(256) void k_pe_sizes(const STask *__restrict__ tasks, int ntasks, const int *__restrict__ ofs, int nwg, int GS, int dest_bits, bool count_far, const u64 *__restrict__ key,
                                                   int *__restrict__ nrec, int *__restrict__ nchunk, unsigned long long *__restrict__ far_total)
{
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= nwg) return;
    const int t0 = (int)w * GS, t1 = min(ntasks, t0 + GS);
    const u64 *K = key + ofs[t0];
    const long long n = ofs[t1] - ofs[t0];
    int nr = 0, nc = 0;
    pack_chunks_wave(K, n, dest_bits, lane, [&](long long, int cnt, unsigned, bool padded) { nr += padded ? ECHUNK : cnt; nc++; });
    if (lane == 0) { nrec[w] = nr; nchunk[w] = nc; }
    if (count_far) {
        long long own_lo = LLONG_MAX, own_hi = LLONG_MIN;
        for (int t = t0; t < t1; t++) { own_lo = min(own_lo, 16LL * tasks[t].row); own_hi = max(own_hi, 16LL * (tasks[t].row + max(1, tasks[t].nrows))); }
        int far = 0;
        for (long long i = lane; i < n; i += 64) { const long long c = (unsigned)K[i]; far += !(c >= own_lo - 2048 && c < own_hi + 2048); }
        far = wave_sum(far);
        if (lane == 0 && far) atomicAdd(far_total, (unsigned long long)far);
    }
}
// one wavefront per group: its records, chunk bases and (panelled plans) panel offsets — panel_offsets' running maximum (plan_tile_ops.h) as a max-scan over the chunk's lanes
__global__ __launch_bounds__(256) void k_pe_write(int ntasks, const int *__restrict__ ofs, int nwg, int GS, int dest_bits, const u64 *__restrict__ key, const int *__restrict__ src, const unsigned short *__restrict__ dest_q,
                                                   const val_t *__restrict__ cval, const int4 *__restrict__ wg, ERec *__restrict__ rec, unsigned *__restrict__ base, int NP, int panel_shift, int *__restrict__ panel_off)
{
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= nwg) return;
    const int t0 = (int)w * GS, t1 = min(ntasks, t0 + GS);
    const u64 *K = key + ofs[t0]; const int *Q = src + ofs[t0];
    const long long n = ofs[t1] - ofs[t0];
    const int4 g = wg[w];
    ERec *R = rec + g.x; unsigned *B = base + g.z;
    int *off = NP > 1 ? panel_off + (size_t)w * (size_t)(NP + 1) : nullptr;
    long long r = 0, c = 0;
    unsigned cur = 0;   // panel of the record before (wavefront-uniform)
    pack_chunks_wave(K, n, dest_bits, lane, [&](long long i, int cnt, unsigned b, bool padded) {
        unsigned wv = 0; bool null_like = true;
        const bool have = lane < cnt || padded;
        if (lane < cnt) {
            const int q = Q[i + lane];
            wv = (((unsigned)K[i + lane] - b) << dest_bits) | dest_q[q];
            const ERec e = make_erec(cval[q], wv);
            R[r + lane] = e;
            null_like = erec_is_null(e);
        } else if (padded) R[r + lane] = make_erec((val_t)0, 0u);
        if (lane == 0) B[c] = b;
        if (off) {
            unsigned e = have && !(null_like && lane != 0) ? (b + (wv >> dest_bits)) >> panel_shift : 0u;   // (0 leaves the running maximum alone)
            for (int d = 1; d < 64; d <<= 1) { const unsigned up = __shfl_up(e, d, 64); if (lane >= d) e = max(e, up); }
            const unsigned pnl = max(cur, e);
            unsigned prev = __shfl_up(pnl, 1, 64);
            if (lane == 0) prev = cur;
            if (have) for (unsigned p = prev + 1; p <= pnl; p++) off[p] = g.x + (int)(r + lane);   // absolute record indices
            cur = __shfl(pnl, 63, 64);
        }
        r += padded ? ECHUNK : cnt; c++;
    });
    if (off && lane == 0) {
        off[0] = g.x;
        for (unsigned p = cur + 1; p <= (unsigned)NP; p++) off[p] = g.x + (int)r;
    }
}


// ---- pooled dictionary plans: the distinct 16-byte patterns (column nibbles, row nibbles) of the packed units, by an open-addressing table in global memory (a shard that qualifies has
// at most 2^DICT_MAX_BITS of them; the table is eight times that).  Nobody ever waits for anybody: a slot is claimed by a 64-bit tag (a hash of the pattern, never 0) with one
// compare-and-swap, the claimant writes the pattern behind it, a unit that meets its own tag takes the slot for its pattern.  That two different patterns share a tag (2^-64 per pair)
// is not assumed away: after the insert kernel has finished, k_pd_pool_verify compares every unit's pattern with the one stored under its tag — a mismatch gives the dictionary up.
constexpr int PTABLE = 8 << DICT_MAX_BITS;
__device__ __forceinline__ u64 pattern_tag(const uint4 q)
{
    u64 h = 1469598103934665603ull;
    for (unsigned w : {q.x, q.y, q.z, q.w}) { h ^= w; h *= 1099511628211ull; h ^= h >> 29; }
    return h ? h : 1ull;
}
__global__ __launch_bounds__(256) void k_pd_pool_patterns(const UDesc *__restrict__ packed, const URow *__restrict__ prow, long long n, int cap, uint4 *__restrict__ table, u64 *__restrict__ tags,
                                                            int *__restrict__ count_over /* [0] distinct so far, [1] over */)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 q = make_uint4(packed[i].n0, packed[i].n1, prow[i].r0, prow[i].r1);
    if (i > 0 && packed[i - 1].n0 == q.x && packed[i - 1].n1 == q.y && prow[i - 1].r0 == q.z && prow[i - 1].r1 == q.w) return;   // (the unit before inserts it)
    const u64 tag = pattern_tag(q);
    unsigned h = (unsigned)(tag >> 20) & (unsigned)(PTABLE - 1);
    for (int probe = 0; probe < PTABLE; probe++, h = (h + 1) & (unsigned)(PTABLE - 1)) {   // (bounded: every pass either ends the walk or moves to the next slot)
        const u64 seen = atomicCAS(&tags[h], 0ull, tag);
        if (seen == 0ull) {   // claimed: this thread alone writes the pattern
            table[h] = q;
            if (atomicAdd(&count_over[0], 1) + 1 > cap) count_over[1] = 1;
            return;
        }
        if (seen == tag) return;
    }
    count_over[1] = 1;   // table full: far more patterns than a dictionary holds
}
__global__ __launch_bounds__(256) void k_pd_pool_verify(const UDesc *__restrict__ packed, const URow *__restrict__ prow, long long n, const uint4 *__restrict__ table, const u64 *__restrict__ tags, int *__restrict__ count_over)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 q = make_uint4(packed[i].n0, packed[i].n1, prow[i].r0, prow[i].r1);
    const u64 tag = pattern_tag(q);
    unsigned h = (unsigned)(tag >> 20) & (unsigned)(PTABLE - 1);
    for (int probe = 0; probe < PTABLE; probe++, h = (h + 1) & (unsigned)(PTABLE - 1)) {
        const u64 seen = tags[h];
        if (seen == tag) { const uint4 t = table[h]; if (t.x != q.x || t.y != q.y || t.z != q.z || t.w != q.w) count_over[1] = 1; return; }
        if (seen == 0ull) break;
    }
    count_over[1] = 1;   // (not found: cannot happen after a complete insert pass; treated like a mismatch)
}
__global__ __launch_bounds__(256) void k_pd_pool_compact(const UDesc *__restrict__ packed, const URow *__restrict__ prow, long long n, const uint4 *__restrict__ dict, int ndict, int word_bits, void *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned q[4] = {packed[i].n0, packed[i].n1, prow[i].r0, prow[i].r1};
    int lo = 0, hi = ndict;   // lower bound in the ascending (n0, n1, r0, r1) dictionary
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint4 d = dict[mid];
        const unsigned e[4] = {d.x, d.y, d.z, d.w};
        bool less = false;
        for (int k = 0; k < 4; k++) { if (e[k] != q[k]) { less = e[k] < q[k]; break; } }
        if (less) lo = mid + 1; else hi = mid;
    }
    const unsigned w0 = packed[i].w0;
    if (word_bits > 0) reinterpret_cast<unsigned *>(out)[i] = (w0 & POOL_BASE_MASK) | ((unsigned)lo << word_bits) | ((w0 >> POOL_KR_SHIFT) << POOL_WORD_KR_SHIFT);   // (hip_plan.h: 4-byte pooled descriptors)
    else reinterpret_cast<uint2 *>(out)[i] = make_uint2(w0, (unsigned)lo);
}
}  // namespace

void DevLists::release()
{
    for (void *q : {(void *)d_rec, (void *)d_base, (void *)d_panel_off}) if (q) (void)hipFree(q);
    d_rec = nullptr; d_base = nullptr; d_panel_off = nullptr;
}

int dev_entry_lists(const val_t *d_cval, const int *d_ccol, const unsigned char *d_crow, long long NC, const std::vector<STask> &tasks, int GS, int slab_shift, int dest_bits, bool count_scattered,
                    int x_panels, int panel_shift, DevLists *L)
{
    const int ntasks = (int)tasks.size(), nwg = (ntasks + GS - 1) / GS;
    L->release(); L->wg.assign((size_t)nwg, make_int4(0, 0, 0, 0)); L->n_rec = L->n_chunk = L->scattered = 0; L->panel_off.clear();
    if (nwg == 0) return 0;
    std::vector<int> ofs((size_t)ntasks + 1, 0);   // entries in front of task t, tasks in their final order
    { long long run = 0; for (int t = 0; t < ntasks; t++) { ofs[(size_t)t] = (int)run; run += tasks[(size_t)t].coo_end - tasks[(size_t)t].coo_begin; } ofs[(size_t)ntasks] = (int)run; if (run > NC) return -3; }
    const long long NE = ofs[(size_t)ntasks];
    Tmp<STask> d_tasks; Tmp<int> d_ofs, src_a, src_b, d_nrec, d_nchunk; Tmp<u64> key_a, key_b; Tmp<unsigned short> d_dest; Tmp<unsigned long long> d_far; Tmp<int4> d_wg;
    PD_TRY(d_tasks.from(tasks)); PD_TRY(d_ofs.from(ofs));
    PD_TRY(key_a.alloc((size_t)NE, false)); PD_TRY(key_b.alloc((size_t)NE, false)); PD_TRY(src_a.alloc((size_t)NE, false)); PD_TRY(src_b.alloc((size_t)NE, false));
    PD_TRY(d_dest.alloc((size_t)NC, true)); PD_TRY(d_nrec.alloc((size_t)nwg, true)); PD_TRY(d_nchunk.alloc((size_t)nwg, true)); PD_TRY(d_far.alloc(1, true));
    hipLaunchKernelGGL(k_pe_keys, dim3((unsigned)std::min(ntasks, 1 << 20)), dim3(256), 0, 0, (const STask *)d_tasks.p, ntasks, (const int *)d_ofs.p, GS, slab_shift, d_ccol, d_crow, key_a.p, src_a.p, d_dest.p);
    PD_TRY(hipGetLastError());
    const u64 *K = key_a.p; const int *Q = src_a.p;
    if (NE > 0) {
        u64 *k_cur = key_a.p, *k_alt = key_b.p; int *v_cur = src_a.p, *v_alt = src_b.p;
        int gbits = 1; while ((1ll << gbits) < nwg) gbits++;
        size_t tmp_b = 0; void *tmp = nullptr;
        PD_TRY(prims::sort_pairs_u64_int(nullptr, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)NE, 0u, (unsigned)(32 + gbits), (hipStream_t)0));
        PD_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = prims::sort_pairs_u64_int(tmp, tmp_b, k_cur, k_alt, v_cur, v_alt, (size_t)NE, 0u, (unsigned)(32 + gbits), (hipStream_t)0);
        (void)hipDeviceSynchronize();
        (void)hipFree(tmp);
        PD_TRY(e);
        K = k_cur; Q = v_cur;
    }
    hipLaunchKernelGGL(k_pe_sizes, dim3(nblk((long long)nwg * 64, 256)), dim3(256), 0, 0, (const STask *)d_tasks.p, ntasks, (const int *)d_ofs.p, nwg, GS, dest_bits, count_scattered, K, d_nrec.p, d_nchunk.p, d_far.p);
    PD_TRY(hipGetLastError());
    std::vector<int> nrec((size_t)nwg), nchunk((size_t)nwg);
    unsigned long long far = 0;
    PD_TRY(hipMemcpy(nrec.data(), d_nrec.p, (size_t)nwg * sizeof(int), hipMemcpyDeviceToHost));
    PD_TRY(hipMemcpy(nchunk.data(), d_nchunk.p, (size_t)nwg * sizeof(int), hipMemcpyDeviceToHost));
    PD_TRY(hipMemcpy(&far, d_far.p, sizeof(far), hipMemcpyDeviceToHost));
    L->scattered = (long long)far;
    long long n_rec = 0, n_chunk = 0;
    for (int w = 0; w < nwg; w++) {
        L->wg[(size_t)w] = make_int4((int)n_rec, (int)(n_rec + nrec[(size_t)w]), (int)n_chunk, 0);
        n_rec += nrec[(size_t)w]; n_chunk += nchunk[(size_t)w];
    }
    L->n_rec = n_rec; L->n_chunk = n_chunk;
    if (n_rec > INT32_MAX) return 0;   // (the caller reports it, like the host builder)
    PD_TRY(d_wg.from(L->wg));
    PD_TRY(hipMalloc((void **)&L->d_rec, std::max<long long>(n_rec, 1) * sizeof(ERec) + 256));
    PD_TRY(hipMalloc((void **)&L->d_base, std::max<long long>(n_chunk, 1) * sizeof(unsigned) + 256));
    const int NP = x_panels;
    if (NP > 1) PD_TRY(hipMalloc((void **)&L->d_panel_off, (size_t)nwg * (size_t)(NP + 1) * sizeof(int) + 256));
    hipLaunchKernelGGL(k_pe_write, dim3(nblk((long long)nwg * 64, 256)), dim3(256), 0, 0, ntasks, (const int *)d_ofs.p, nwg, GS, dest_bits, K, Q, (const unsigned short *)d_dest.p, d_cval, (const int4 *)d_wg.p, L->d_rec, L->d_base, NP,
                       panel_shift, L->d_panel_off);
    PD_TRY(hipGetLastError());
    PD_TRY(hipDeviceSynchronize());
    if (NP > 1) {
        L->panel_off.resize((size_t)nwg * (size_t)(NP + 1));
        PD_TRY(hipMemcpy(L->panel_off.data(), L->d_panel_off, L->panel_off.size() * sizeof(int), hipMemcpyDeviceToHost));
    }
    return 0;
}

void DevCounts::release()
{
    for (void *q : {(void *)tu, (void *)pool}) if (q) (void)hipFree(q);   // (tu heads the one block that holds tc, td, tp, pool_u, pool_c too)
    tu = tc = td = tp = pool_u = pool_c = nullptr; pool = nullptr; csr_form = -1;
}

int dev_fetch_ints(const int *d_array, const long long *idx, int n, int *out)
{
    if (n <= 0) return 0;
    Tmp<long long> d_idx; Tmp<int> d_out;
    PD_TRY(d_idx.alloc((size_t)n, false)); PD_TRY(d_out.alloc((size_t)n, false));
    PD_TRY(hipMemcpy(d_idx.p, idx, (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pd_gather_ints, dim3(nblk(n, 256)), dim3(256), 0, 0, d_array, (const long long *)d_idx.p, n, d_out.p);
    PD_TRY(hipGetLastError());
    PD_TRY(hipMemcpy(out, d_out.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

int dev_count(const DevShard &S, int csr_form, DevCounts *C, hvec<int> &counts3, long long *pool_units, long long *pool_lines)
{
    const DevTile *D = S.D;
    const int nt = S.t_end - S.t_begin, ntr = S.tr1 - S.tr0;
    const bool verbose = getenv("TILESPMV_PLAN_VERBOSE") != nullptr;
    timeval tv0; gettimeofday(&tv0, NULL);
    auto lap_ms = [&]() { timeval t; gettimeofday(&t, NULL); const double ms = (t.tv_sec - tv0.tv_sec) * 1e3 + (t.tv_usec - tv0.tv_usec) * 1e-3; tv0 = t; return ms; };
    C->release();
    C->csr_form = csr_form;
    counts3.resize((size_t)ntr * 3);
    // one allocation for the per-tile prefix arrays (and the pooled counts): tu | tc | td | tp | pool_u | pool_c
    const size_t per = ((size_t)nt + 1 + 63) / 64 * 64, perr = ((size_t)std::max(ntr, 1) + 63) / 64 * 64;
    int *blockp = nullptr;
    const size_t ints = 3 * per + (csr_form >= 2 ? per + 2 * perr : 0) + 16;
    PD_TRY(hipMalloc((void **)&blockp, ints * sizeof(int)));
    C->tu = blockp; C->tc = blockp + per; C->td = blockp + 2 * per;
    if (csr_form >= 2) C->tp = blockp + 3 * per;
    PD_TRY(hipMemsetAsync(blockp, 0, ints * sizeof(int), 0));
    if (nt > 0) {
        hipLaunchKernelGGL(k_pd_count_tiles, dim3(nblk(nt, 256)), dim3(256), 0, 0, D->T, D->tile_bi, D->hyb_off, S.t_begin, nt, D->rowA, D->colA, S.coo_in_tile, S.dense_mfma, csr_form, S.absorb && csr_form < 2, C->tu, C->tc, C->td, C->tp);
        PD_TRY(hipGetLastError());
        size_t tmp_b = 0; void *tmp = nullptr;
        PD_TRY(prims::scan_int(nullptr, tmp_b, C->tu, C->tu, (size_t)nt + 1, (hipStream_t)0));
        PD_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = hipSuccess;
        for (int *a : {C->tu, C->tc, C->td, C->tp}) if (a && e == hipSuccess) e = prims::scan_int(tmp, tmp_b, a, a, (size_t)nt + 1, (hipStream_t)0);   // (same stream: the scans run one after the other)
        if (e == hipSuccess) e = hipDeviceSynchronize();
        (void)hipFree(tmp);
        PD_TRY(e);
    }
    const double ms_tiles = lap_ms();
    if (csr_form >= 2 && ntr > 0) {
        C->pool_u = blockp + 4 * per; C->pool_c = blockp + 4 * per + perr;
        PD_TRY(hipMalloc((void **)&C->pool, (size_t)std::max<long long>(1, S.stored) * sizeof(PoolEnt) + 16));
        Tmp<unsigned long long> d_stat;
        PD_TRY(d_stat.alloc(2, true));
        if (nt > 0) { hipLaunchKernelGGL(k_pd_fill_pool, dim3(nblk(nt, 256)), dim3(256), 0, 0, D->T, D->tile_bi, D->hyb_off, S.t_begin, nt, D->rowA, S.coo_in_tile, (const int *)C->tp, C->pool); PD_TRY(hipGetLastError()); }
        hipLaunchKernelGGL(k_pd_count_pool, dim3(nblk((long long)ntr * 64, 256)), dim3(256), 0, 0, (const int *)D->T.tile_ptr, S.tr0, ntr, S.t_begin, csr_form == 3 ? POOL_WIDE_WINDOW : 16u, (const int *)C->tp, (const PoolEnt *)C->pool, C->pool_u, C->pool_c, d_stat.p);
        PD_TRY(hipGetLastError());
        unsigned long long h_stat[2] = {0, 0};
        PD_TRY(hipMemcpy(h_stat, d_stat.p, sizeof(h_stat), hipMemcpyDeviceToHost));
        if (pool_units) *pool_units += (long long)h_stat[0];
        if (pool_lines) *pool_lines += (long long)h_stat[1];
    }
    if (ntr > 0) {
        Tmp<int> d_out;
        PD_TRY(d_out.alloc((size_t)ntr * 3, false));
        hipLaunchKernelGGL(k_pd_row_counts, dim3(nblk(ntr, 256)), dim3(256), 0, 0, (const int *)D->T.tile_ptr, S.tr0, ntr, S.t_begin, (const int *)C->tu, (const int *)C->tc, (const int *)C->td,
                           (const int *)C->pool_u, (const int *)C->pool_c, d_out.p);
        PD_TRY(hipGetLastError());
        PD_TRY(hipMemcpy(counts3.data(), d_out.p, counts3.size() * sizeof(int), hipMemcpyDeviceToHost));
    }
    if (verbose) fprintf(stderr, "tilespmv: device count (form %d): per-tile counts + scans %.1f ms, pooled windows + per-row counts to the host %.1f ms\n", csr_form, ms_tiles, lap_ms());
    return 0;
}

int dev_pattern_sample(const DevShard &S, int step, int last_first_tile, int last_rowlen, std::vector<unsigned long long> &patterns)
{
    patterns.clear();
    const int nt = S.t_end - S.t_begin;
    if (nt <= 0) return 0;
    const int nsample = (nt + step - 1) / step;
    Tmp<u64> d_p; Tmp<int> d_n;
    PD_TRY(d_p.alloc((size_t)nsample * 16, false)); PD_TRY(d_n.alloc((size_t)nsample, true));
    hipLaunchKernelGGL(k_pd_pattern_sample, dim3(nblk(nsample, 256)), dim3(256), 0, 0, S.D->T, S.t_begin, S.t_end, step, nsample, last_first_tile, last_rowlen, d_p.p, d_n.p);
    PD_TRY(hipGetLastError());
    std::vector<u64> hp((size_t)nsample * 16); std::vector<int> hn((size_t)nsample);
    PD_TRY(hipMemcpy(hp.data(), d_p.p, hp.size() * sizeof(u64), hipMemcpyDeviceToHost));
    PD_TRY(hipMemcpy(hn.data(), d_n.p, hn.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int s = 0; s < nsample; s++) for (int q = 0; q < hn[(size_t)s]; q++) patterns.push_back(hp[(size_t)s * 16 + q]);
    return 0;
}

int dev_emit(const DevShard &S, const DevCounts &C, const hvec<long long> &pu, const hvec<long long> &pc, const hvec<long long> &pd, const std::vector<unsigned char> &row_k,
             const std::vector<unsigned char> &row_split, long long NU, const EmitOut &O)
{
    const DevTile *D = S.D;
    const int nt = S.t_end - S.t_begin, ntr = S.tr1 - S.tr0;
    if (ntr <= 0) return 0;
    auto narrow = [](const hvec<long long> &v) { std::vector<int> o(v.size()); for (size_t i = 0; i < v.size(); i++) o[i] = (int)v[i]; return o; };   // (the builder refuses shards beyond 2^31 units / entries)
    Tmp<int> d_pu, d_pc, d_pd; Tmp<unsigned char> d_rk, d_rs;
    PD_TRY(d_pu.from(narrow(pu))); PD_TRY(d_pc.from(narrow(pc))); PD_TRY(d_pd.from(narrow(pd)));
    PD_TRY(d_rk.from(row_k)); PD_TRY(d_rs.from(row_split));
    if (O.urow && NU > 0) { hipLaunchKernelGGL(k_pd_fill_urow, dim3(nblk(NU, 256)), dim3(256), 0, 0, O.urow, NU); PD_TRY(hipGetLastError()); }
    if (nt > 0) {
        hipLaunchKernelGGL(k_pd_emit_tiles, dim3(nblk(nt, 256)), dim3(256), 0, 0, D->T, D->tile_bi, D->hyb_off, S.t_begin, nt, S.tr0, D->rowA, D->colA, S.coo_in_tile, S.dense_mfma, C.csr_form, S.absorb && C.csr_form < 2, S.derive && C.csr_form < 2, (const int *)C.tu,
                           (const int *)C.tc, (const int *)C.td, (const int *)d_pu.p, (const int *)d_pc.p, (const int *)d_pd.p, (const unsigned char *)d_rk.p, (const unsigned char *)d_rs.p, O);
        PD_TRY(hipGetLastError());
    }
    if (C.csr_form >= 2) {
        hipLaunchKernelGGL(k_pd_emit_pool, dim3(nblk((long long)ntr * 64, 256)), dim3(256), 0, 0, (const int *)D->T.tile_ptr, S.tr0, ntr, S.t_begin, C.csr_form == 3 ? POOL_WIDE_WINDOW : 16u, (const int *)C.tp, (const PoolEnt *)C.pool, (const int *)C.tu, (const int *)C.tc, (const int *)d_pu.p,
                           (const int *)d_pc.p, (const unsigned char *)d_rk.p, O);
        PD_TRY(hipGetLastError());
    }
    PD_TRY(hipDeviceSynchronize());
    return 0;
}

int dev_fetch_word0(const uint4 *d_udesc, long long NU, hvec<unsigned> &w0)
{
    w0.assign((size_t)NU, 0u);
    if (NU <= 0) return 0;
    Tmp<unsigned> d;
    PD_TRY(d.alloc((size_t)NU, false));
    hipLaunchKernelGGL(k_pd_word0, dim3(nblk(NU, 256)), dim3(256), 0, 0, d_udesc, NU, d.p);
    PD_TRY(hipGetLastError());
    PD_TRY(hipMemcpy(w0.data(), d.p, (size_t)NU * sizeof(unsigned), hipMemcpyDeviceToHost));
    return 0;
}

int dev_pack_desc(const uint4 *d_udesc, const uint2 *d_urow, const uint4 *d_ucol, const int4 *d_map, int ntasks, UDesc *d_packed, URow *d_packed_row, uint4 *d_packed_col)
{
    if (ntasks <= 0) return 0;
    hipLaunchKernelGGL(k_pd_pack_desc, dim3((unsigned)std::min(ntasks, 1 << 20)), dim3(256), 0, 0, d_udesc, d_urow, d_ucol, d_map, ntasks, d_packed, d_packed_row, d_packed_col);
    PD_TRY(hipGetLastError());
    PD_TRY(hipDeviceSynchronize());
    return 0;
}

int dev_shift_histogram(const UDesc *d_packed, long long NUP, unsigned long long hist[8])
{
    for (int c = 0; c < 8; c++) hist[c] = 0;
    if (NUP <= 0) return 0;
    Tmp<unsigned long long> h;
    PD_TRY(h.alloc(8, true));
    hipLaunchKernelGGL(k_pd_shift_hist, dim3((unsigned)std::min<long long>(nblk(NUP, 256), 4096)), dim3(256), 0, 0, d_packed, NUP, h.p);
    PD_TRY(hipGetLastError());
    PD_TRY(hipMemcpy(hist, h.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

int dev_dict_patterns(const UDesc *d_packed, long long NUP, size_t cap, std::vector<uint4> &dict, DictRanges *ranges, bool *over)
{
    dict.clear(); *over = false;
    for (int c = 0; c < 9; c++) ranges->off[c] = 0;
    if (NUP <= 0) return 0;
    // how many units carry each shift code: one sort + run-length pass per code that occurs (plans without absorbed entries: code 0 only, the whole array as before)
    Tmp<unsigned long long> hist;
    PD_TRY(hist.alloc(9, true));
    hipLaunchKernelGGL(k_pd_shift_hist, dim3((unsigned)std::min<long long>(nblk(NUP, 256), 4096)), dim3(256), 0, 0, d_packed, NUP, hist.p);
    PD_TRY(hipGetLastError());
    unsigned long long h[9];
    PD_TRY(hipMemcpy(h, hist.p, sizeof(h), hipMemcpyDeviceToHost));
    Tmp<u64> a, b, uniq; Tmp<int> counts, nruns;
    PD_TRY(a.alloc((size_t)NUP, false)); PD_TRY(b.alloc((size_t)NUP, false));
    PD_TRY(uniq.alloc((size_t)NUP, false)); PD_TRY(counts.alloc((size_t)NUP, false)); PD_TRY(nruns.alloc(1, true));
    std::vector<u64> got;
    for (unsigned code = 0; code < 8; code++) {
        ranges->off[code] = (int)dict.size();
        const long long nc = (long long)h[code];
        if (nc == 0) continue;
        PD_TRY(hipMemset(hist.p + 8, 0, sizeof(unsigned long long)));
        hipLaunchKernelGGL(k_pd_patterns, dim3(nblk(NUP, 256)), dim3(256), 0, 0, d_packed, NUP, code, nc == NUP, a.p, hist.p + 8);
        PD_TRY(hipGetLastError());
        u64 *k_cur = a.p, *k_alt = b.p;
        size_t tmp_b = 0; void *tmp = nullptr;
        PD_TRY(prims::sort_keys_u64(nullptr, tmp_b, k_cur, k_alt, (size_t)nc, 0u, 64u, (hipStream_t)0));
        PD_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        hipError_t e = prims::sort_keys_u64(tmp, tmp_b, k_cur, k_alt, (size_t)nc, 0u, 64u, (hipStream_t)0);
        (void)hipDeviceSynchronize();
        (void)hipFree(tmp);
        PD_TRY(e);
        PD_TRY(hipMemset(nruns.p, 0, sizeof(int)));
        tmp_b = 0; tmp = nullptr;
        PD_TRY(prims::rle_u64(nullptr, tmp_b, k_cur, 0u, (unsigned)nc, uniq.p, counts.p, nruns.p, (hipStream_t)0));
        PD_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_b, 16)));
        e = prims::rle_u64(tmp, tmp_b, k_cur, 0u, (unsigned)nc, uniq.p, counts.p, nruns.p, (hipStream_t)0);
        int n = 0;
        if (e == hipSuccess) e = hipMemcpy(&n, nruns.p, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        PD_TRY(e);
        if (dict.size() + (size_t)n > cap) { *over = true; dict.clear(); return 0; }
        got.resize((size_t)n);
        PD_TRY(hipMemcpy(got.data(), uniq.p, (size_t)n * sizeof(u64), hipMemcpyDeviceToHost));
        for (u64 k : got) dict.push_back(make_uint4((unsigned)(k >> 32), (unsigned)(k & 0xffffffffull), code << UNIT_SHIFT_SHIFT, 0u));
    }
    ranges->off[8] = (int)dict.size();
    return 0;
}

int dev_pool_dict(const UDesc *d_packed, const URow *d_packed_row, long long NUP, size_t cap, std::vector<uint4> &sorted_patterns, bool *over)
{
    sorted_patterns.clear(); *over = false;
    if (NUP <= 0) return 0;
    Tmp<uint4> table; Tmp<u64> tags; Tmp<int> cnt;
    PD_TRY(table.alloc((size_t)PTABLE, true)); PD_TRY(tags.alloc((size_t)PTABLE, true)); PD_TRY(cnt.alloc(2, true));
    hipLaunchKernelGGL(k_pd_pool_patterns, dim3(nblk(NUP, 256)), dim3(256), 0, 0, d_packed, d_packed_row, NUP, (int)cap, table.p, tags.p, cnt.p);
    PD_TRY(hipGetLastError());
    int h_cnt[2] = {0, 0};
    PD_TRY(hipMemcpy(h_cnt, cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost));
    if (h_cnt[1] || (size_t)h_cnt[0] > cap) { *over = true; return 0; }
    hipLaunchKernelGGL(k_pd_pool_verify, dim3(nblk(NUP, 256)), dim3(256), 0, 0, d_packed, d_packed_row, NUP, (const uint4 *)table.p, (const u64 *)tags.p, cnt.p);
    PD_TRY(hipGetLastError());
    PD_TRY(hipMemcpy(h_cnt, cnt.p, sizeof(h_cnt), hipMemcpyDeviceToHost));
    if (h_cnt[1]) { *over = true; return 0; }   // (two patterns under one tag: the 20-byte form stays)
    std::vector<uint4> ht((size_t)PTABLE); std::vector<u64> hs((size_t)PTABLE);
    PD_TRY(hipMemcpy(ht.data(), table.p, ht.size() * sizeof(uint4), hipMemcpyDeviceToHost));
    PD_TRY(hipMemcpy(hs.data(), tags.p, hs.size() * sizeof(u64), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ht.size(); i++) if (hs[i] != 0ull) sorted_patterns.push_back(ht[i]);
    std::sort(sorted_patterns.begin(), sorted_patterns.end(), [](const uint4 &a, const uint4 &b) {
        if (a.x != b.x) return a.x < b.x;
        if (a.y != b.y) return a.y < b.y;
        if (a.z != b.z) return a.z < b.z;
        return a.w < b.w;
    });
    return 0;
}

int dev_pool_compact(const UDesc *d_packed, const URow *d_packed_row, long long NUP, const uint4 *d_dict, int ndict, int word_bits, void *d_out)
{
    if (NUP <= 0) return 0;
    hipLaunchKernelGGL(k_pd_pool_compact, dim3(nblk(NUP, 256)), dim3(256), 0, 0, d_packed, d_packed_row, NUP, d_dict, ndict, word_bits, d_out);
    PD_TRY(hipGetLastError());
    PD_TRY(hipDeviceSynchronize());
    return 0;
}

int dev_compact_desc(const UDesc *d_packed, long long NUP, const uint4 *d_dict, DictRanges ranges, int cb_bits, unsigned *d_compact)
{
    if (NUP <= 0) return 0;
    hipLaunchKernelGGL(k_pd_compact, dim3(nblk(NUP, 256)), dim3(256), 0, 0, d_packed, NUP, d_dict, ranges, cb_bits, d_compact);
    PD_TRY(hipGetLastError());
    PD_TRY(hipDeviceSynchronize());
    return 0;
}

}  // namespace tilespmv
