// hip_kernels.hip — hand-written CDNA4 (gfx950, wave64) kernels of the y = A*x hot path.
//
//   k_units          (default) unit-stream kernel: one 16-lane strip per task consumes a flat run of
//                    self-describing 16-value units (ELL / HYB slots, dense and dense-col columns, the
//                    leading entries of CSR tile rows, dense-row "row units") plus the COO entry list —
//                    walked per strip, or, on entry-heavy shards, as one merged, column-ordered list per
//                    wavefront / per workgroup.  Replaces stir_spmv_cuda_kernel_v6 (src/tilespmv_cuda.h:394-792).
//   k_dense_mfma     dense tiles on the matrix cores, one wavefront per tile-row (y +=).
//   k_tiles_direct   first-generation fused tile SpMV: a strip walks whole tile-rows one tile at a
//                    time through the seven per-tile routines below (TILESPMV_KERNEL=1), and the
//                    pass for CSR tiles kept whole in generation 2 (ACCUM).
//   k_fixup_split    sums the partial results of split (very long) tile-rows in a fixed order
//                    (the reference uses global atomicAdd, src/tilespmv_cuda.h:784-790).
//   k_fallback_entries  very-sparse fallback, y += A_coo x over the extracted matrix: nnz-balanced row blocks,
//                    column-ordered, one workgroup each (the reference hands this to CSR5, src/tilespmv_cuda.h:1011-1029,:1080).
//
// Per-tile routines (lane r = lane & 15 owns row r of the tile):
//   CSR      reference src/tilespmv_cuda.h:531-561   row-per-lane walk of the byte row pointer
//   COO      :462-488                                16 entries per step, LDS scatter-add
//   ELL      :579-605                                slot-major, 128 B of values per step
//   HYB      :606-663 (+ v5 :185-234 remainder)      ELL part + COO remainder
//   dense    :664-710                                v_mfma_f64_16x16x4_f64 (wave-cooperative) or VALU
//   dense-row:711-750                                16-lane DPP reduction per dense row
//   dense-col:751-778                                like ELL with a per-column id
// x is staged per tile as a 16-value LDS segment (the reference: s_x_warp / register + shfl).
#include <hip/hip_runtime.h>

#include "hip_plan.h"

// Timing-only diagnostics (ablations whose results are wrong by construction, clock stamps, cache-policy probes) live in hip_kernels_diag.h, which only a diagnostic build
// includes: `make VARIANT=_name EXTRA_DEFS=-D...` adds -DTILESPMV_DIAG and writes lib*_name.so; the Makefile refuses EXTRA_DEFS without VARIANT, so no stray define can
// turn the product libraries into something that returns wrong rows.  In a product build every hook below expands to nothing.
#if !defined(TILESPMV_DIAG) && (defined(TILESPMV_ABL) || defined(XCD_ABL) || defined(TILESPMV_STAMPS) || defined(TILESPMV_POOL_ABL) || defined(TILESPMV_ABL_LDS_PAD) || defined(TILESPMV_GATHER_POLICY))
#error "diagnostic defines (TILESPMV_ABL, XCD_ABL, TILESPMV_STAMPS, TILESPMV_POOL_ABL, TILESPMV_ABL_LDS_PAD, TILESPMV_GATHER_POLICY) need -DTILESPMV_DIAG: build with make VARIANT=_name EXTRA_DEFS=..."
#endif
#ifdef TILESPMV_DIAG
#include "hip_kernels_diag.h"
#else
#define TSPMV_DIAG_GATHER_X(p) (*(p))
#define TSPMV_DIAG_TRIP_DECL
#define TSPMV_DIAG_TRIP_RECORD(r, q, e0)
#define TSPMV_DIAG_TRIP_GATHERS
#define TSPMV_DIAG_TRIP_ADDS_REPLACED 0
#define TSPMV_DIAG_TRIP_ADDS
#define TSPMV_DIAG_UNITS_LDS_PAD
#define TSPMV_DIAG_UNIT_GATHER_SKIP(k) false
#define TSPMV_DIAG_UNIT_X(i) (i)
#define TSPMV_DIAG_COO0_ON true
#define TSPMV_DIAG_COO0_X(c) (c)
#define TSPMV_DIAG_POOL_ADD(dest, prod) false
#define TSPMV_DIAG_POOL_X(load, d) (load)
#define TSPMV_DIAG_XCD_ZERO 1
#define TSPMV_DIAG_XCD_TRIP 1
#define TSPMV_DIAG_XCD_SKIP_ADDS
#define TSPMV_STAMP_DECL
#define TSPMV_STAMP(i) do { } while (0)
#define TSPMV_STAMP_WAIT(i) do { } while (0)
#define TSPMV_STAMP_STORE
#endif

namespace tilespmv {

// LDS scatter accumulators are fp64 in BOTH builds: on gfx950 a wavefront's ds_add_f32 takes about 170 cycles whatever the address
// pattern, its ds_add_f64 7-18 (scripts/micro/lds_atomic_rate.hip: 203 against 1860-4670 G adds/s) — the fp32 build's entry phase
// spent 0.108 ms of a 0.166 ms SpMV (power-law 8 M rows) in them.  Products are formed in the value type and widened for the add;
// sums of integer-valued data stay exact, real-valued sums get closer to the exact result than a float chain would.
typedef double lacc_t;


typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int GROUPS_PER_BLOCK = 16;  // 256 threads
#ifndef TILESPMV_STREAM_MIN_WAVES
#define TILESPMV_STREAM_MIN_WAVES 1     // second __launch_bounds__ argument = waves per SIMD asked of the register allocator
#endif

__device__ __forceinline__ int nibble_of(const unsigned char *__restrict__ base, int p)
{
    const unsigned b = base[p >> 1];
    return (p & 1) ? (int)(b & 15u) : (int)(b >> 4);
}

// ---- 16-lane all-reduce with DPP row rotations (a DPP "row" is exactly one 16-lane strip)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <class T>
__device__ __forceinline__ T strip_allreduce(T v)
{
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    return v;
}

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wavefront complete in issue order; this only stops the compiler
    // from moving accesses across the point where other lanes' data is consumed.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- dense tile on the matrix cores, whole wavefront on one tile:
// A[row][k] = tile[row][4s+k] (64 consecutive values = one coalesced 512-B load per k-step),
// B[k][*]  = x[4s+k];  after 4 steps D[row][*] = (tile * x)[row] in every column.
__device__ __forceinline__ void mfma_dense_tile(const val_t *__restrict__ tile, const val_t *xseg, int lane, val_t *out16)
{
#if defined(TILESPMV_F32)
    v4f d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; s++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(tile[64 * s + lane], xseg[4 * s + (lane >> 4)], d, 0, 0, 0);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) out16[4 * (lane >> 4) + i] = d[i];  // f32 C/D map: row = 4*(lane>>4)+i
    }
#else
    v4d d = {0., 0., 0., 0.};
#pragma unroll
    for (int s = 0; s < 4; s++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(tile[64 * s + lane], xseg[4 * s + (lane >> 4)], d, 0, 0, 0);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) out16[(lane >> 4) + 4 * i] = d[i];  // f64 C/D map: row = (lane>>4)+4*i
    }
#endif
}

// ================================================================================================
// Fused tile kernel, direct global loads.
// ================================================================================================
template <bool DENSE_MFMA, bool ACCUM>
__global__ __launch_bounds__(256) void k_tiles_direct(DevPlan P, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    __shared__ val_t s_x[GROUPS_PER_BLOCK][16];    // x segment of the strip's current tile
    __shared__ lacc_t s_acc[GROUPS_PER_BLOCK][16]; // scatter accumulator (COO / HYB remainder)
    __shared__ val_t s_t[4][16];                   // MFMA result hand-off, one per wave

    const int tid = threadIdx.x, lane = tid & 63, r = tid & 15, g = tid >> 4, wave = tid >> 6;
    const long long task_id = (long long)blockIdx.x * GROUPS_PER_BLOCK + g;
    const bool have = task_id < P.ntasks;
    Task tk;
    if (have) tk = P.task[task_id];
    else { tk.tile_begin = tk.tile_end = 0; tk.val_off = tk.idx_off = 0; tk.row = 0; tk.partial = -1; }

    int t = tk.tile_begin;
    const int tend = tk.tile_end;
    long long voff = tk.val_off, ioff = tk.idx_off;
    int row = tk.row;
    val_t acc = 0;
    bool dirty = false;
    s_acc[g][r] = 0;

    while (__ballot(t < tend) != 0ull) {
        const bool on = t < tend;
        unsigned meta = DESC_FMT_NOP;
        int cb = 0;
        if (on) {
            const uint2 d = P.desc[t];
            cb = (int)d.x; meta = d.y;
            const long long xi = (long long)cb * 16 + r;
            s_x[g][r] = (xi < P.colA) ? x[xi] : (val_t)0;
        }
        const int fmt = (int)(meta & DESC_FMT_MASK);
        const int p1 = (int)((meta >> DESC_P1_SHIFT) & 255u), p2 = (int)((meta >> DESC_P2_SHIFT) & 255u);
        wave_lds_fence();

        if (DENSE_MFMA) {  // wave-uniform: every strip of the wave helps with each dense tile in turn
            unsigned long long pending = __ballot(on && fmt == TILESPMV_FMT_DNS);
            while (pending) {
                const int gl = (__ffsll((long long)pending) - 1) >> 4;  // strip (0..3) inside the wave
                const int lo = __builtin_amdgcn_readlane((int)(voff & 0xffffffffll), gl * 16);
                const int hi = __builtin_amdgcn_readlane((int)(voff >> 32), gl * 16);
                const long long vo = ((long long)hi << 32) | (unsigned)lo;
                mfma_dense_tile(P.val + vo, &s_x[wave * 4 + gl][0], lane, &s_t[wave][0]);
                wave_lds_fence();
                if (((lane >> 4) == gl)) acc += s_t[wave][r];
                wave_lds_fence();
                pending &= ~(0xFFFFull << (gl * 16));
            }
        }

        if (on) {
            const val_t *__restrict__ v = P.val + voff;
            const unsigned char *__restrict__ ix = P.idx + ioff;
            const val_t *xs = &s_x[g][0];
            int nv = 0, ni = 0;
            switch (fmt) {
            case TILESPMV_FMT_ELL: {
                for (int s = 0; s < p1; s++) acc += v[16 * s + r] * xs[nibble_of(ix, 16 * s + r)];
                nv = 16 * p1; ni = 8 * p1;
                break;
            }
            case TILESPMV_FMT_CSR: {
                const int k0 = ix[r], k1 = (r == 15) ? p1 : (int)ix[r + 1];
                for (int k = k0; k < k1; k++) acc += v[k] * xs[nibble_of(ix + 16, k)];
                nv = p1; ni = 16 + ((p1 + 1) >> 1);
                break;
            }
            case TILESPMV_FMT_COO: {
                if (r < p1) {
                    const unsigned b = ix[r];
                    atomicAdd(&s_acc[g][b >> 4], (lacc_t)(v[r] * xs[b & 15u]));
                }
                dirty = true; nv = p1; ni = p1;
                break;
            }
            case TILESPMV_FMT_HYB: {
                for (int s = 0; s < p1; s++) acc += v[16 * s + r] * xs[nibble_of(ix, 16 * s + r)];
                if (r < p2) {
                    const unsigned b = ix[8 * p1 + r];
                    atomicAdd(&s_acc[g][b >> 4], (lacc_t)(v[16 * p1 + r] * xs[b & 15u]));
                }
                dirty = true; nv = 16 * p1 + p2; ni = 8 * p1 + p2;
                break;
            }
            case TILESPMV_FMT_DNS: {
                if (!DENSE_MFMA) {
#pragma unroll 4
                    for (int c = 0; c < 16; c++) acc += v[16 * c + r] * xs[c];
                }
                nv = 256; ni = 0;
                break;
            }
            case TILESPMV_FMT_DNSROW: {
                const val_t xr = xs[r];
                for (int k = 0; k < p1; k++) {
                    const val_t sum = strip_allreduce(v[16 * k + r] * xr);
                    if (r == (int)ix[k]) acc += sum;
                }
                nv = 16 * p1; ni = p1;
                break;
            }
            case TILESPMV_FMT_DNSCOL: {
                for (int k = 0; k < p1; k++) acc += v[16 * k + r] * xs[ix[k]];
                nv = 16 * p1; ni = p1;
                break;
            }
            default: break;
            }
            voff += nv; ioff += ni;
            t++;
            if ((meta & DESC_EOR) && tk.partial < 0) {  // tile-row finished: write its 16 results
                wave_lds_fence();
                val_t out = acc;
                if (dirty) { out = (val_t)((lacc_t)out + s_acc[g][r]); s_acc[g][r] = 0; dirty = false; }
                const long long yi = (long long)row * 16 + r;
                if (yi < P.rowA) { if (ACCUM) y[yi] += out; else y[yi] = out; }
                acc = 0; row++;
            }
        }
        wave_lds_fence();
    }
    if (have && tk.partial >= 0) {  // piece of a split tile-row: combined later in a fixed order
        val_t out = acc;
        if (dirty) out = (val_t)((lacc_t)out + s_acc[g][r]);
        P.partial[(long long)tk.partial * 16 + r] = out;
    }
}

__global__ __launch_bounds__(256) void k_fixup_split(DevPlan P, val_t *__restrict__ y)
{
    const int r = threadIdx.x & 15;
    const int i = blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x >> 4);
    if (i >= P.nfix) return;
    const FixRow f = P.fix[i];
    val_t sum = 0;
    for (int k = 0; k < f.count; k++) sum += P.partial[(long long)(f.first + k) * 16 + r];
    const long long yi = (long long)f.row * 16 + r;
    if (yi < P.rowA) y[yi] = sum;
}

// ---- launchers (host) ---------------------------------------------------------------------------
hipError_t launch_tiles_direct(const DevPlan &P, bool dense_mfma, bool accumulate, bool fixup, const val_t *x, val_t *y, hipStream_t st)
{
    if (P.ntasks > 0) {
        const dim3 grid((unsigned)((P.ntasks + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK)), blk(256);
        if (dense_mfma) { if (accumulate) hipLaunchKernelGGL((k_tiles_direct<true, true>), grid, blk, 0, st, P, x, y);
                          else hipLaunchKernelGGL((k_tiles_direct<true, false>), grid, blk, 0, st, P, x, y); }
        else { if (accumulate) hipLaunchKernelGGL((k_tiles_direct<false, true>), grid, blk, 0, st, P, x, y);
               else hipLaunchKernelGGL((k_tiles_direct<false, false>), grid, blk, 0, st, P, x, y); }
    }
    if (fixup && P.nfix > 0)
        hipLaunchKernelGGL(k_fixup_split, dim3((P.nfix + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(256), 0, st, P, y);
    return hipGetLastError();
}

}  // namespace tilespmv

// ================================================================================================
// Second generation: unit-stream kernel (layout: hip_plan.h "unit stream").
// One 16-lane strip per task as before, but the common formats are consumed as a flat run of
// self-describing 16-value units whose addresses depend only on the unit index:
//   phase 1  COO entry list of the strip -> LDS scatter-add (ds_add) into s_y[strip row][row]
//   phase 2  units in batches of UB: descriptors come from LDS (one coalesced load per 16 units), the
//            x gathers of a batch are issued at once, the next batch's value loads go in flight before
//            the first use; a finished tile-row parks its 16 results in LDS and y is written once per
//            strip with 16-B lane stores.
// Dense tiles for the matrix cores (k_dense_mfma) and CSR tiles kept whole (k_tiles_direct<.., ACCUM>)
// run after this kernel, which keeps it at 60 VGPRs (8 waves/SIMD).
// ================================================================================================
namespace tilespmv {

template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// 12-B descriptor in HBM -> the 16-B form the strip parks in LDS (both halves carry word 0)
__device__ __forceinline__ uint4 load_udesc(const UDesc *__restrict__ d, int i)
{
    const UDesc u = d[i];
    return make_uint4(u.w0, u.n0, u.w0, u.n1);
}

// ... and the same descriptor as it is held in registers while it waits for its turn (k_units): the three loaded words as they arrive, (w0, n0, n1, -).  Building the LDS form
// right behind the load needs a register move of w0, and the move needs the load's data: the wavefront then waits a full memory round trip for a prefetch it
// will not use for four batches (that is what the 12-byte-descriptor kernels did until round 5).  The LDS form is built when the chunk is parked.
__device__ __forceinline__ uint4 load_udesc_raw(const UDesc *__restrict__ d, int i)
{
    const UDesc u = d[i];
    return make_uint4(u.w0, u.n0, u.n1, 0u);
}
__device__ __forceinline__ uint4 udesc_park_form(const uint4 raw) { return make_uint4(raw.x, raw.y, raw.x, raw.z); }

// Dictionary plans (DevStream::cb_bits > 0; hip_plan.hip): 4 B per unit in HBM — column block | pattern id << cb_bits |
// flags << 27 — and the unit's column pattern (the two nibble words) in a small dictionary that stays in the vector L1.
// A lane expands its unit's descriptor to the 16-B LDS form when the chunk is parked.
__device__ __forceinline__ uint4 udict_of(const DevStream &S, unsigned w) { return S.udict[(w << 5) >> (5 + S.cb_bits)]; }   // (nibbles of rows 0-7, of rows 8-15, window shift << 29, 0)
// pooled dictionary plans: descriptor of unit i as (word 0 = window base | tile-row in strip << POOL_KR_SHIFT, pattern id) — from the 8-byte pair, or (S.cb_bits = b > 0) from the
// 4-byte word base | id << b | tile-row << 30 (hip_plan.h)
__device__ __forceinline__ uint2 pool_desc(const DevStream &S, int i)
{
    const int b = S.cb_bits;
    if (b > 0) {
        const unsigned a = reinterpret_cast<const unsigned *>(S.udesc)[i];
        return make_uint2((a & ((1u << b) - 1u)) | ((a >> POOL_WORD_KR_SHIFT) << POOL_KR_SHIFT), (a << (32 - POOL_WORD_KR_SHIFT)) >> (32 - POOL_WORD_KR_SHIFT + b));
    }
    return reinterpret_cast<const uint2 *>(S.udesc)[i];
}
__device__ __forceinline__ uint4 udesc_expand(const DevStream &S, unsigned w, uint4 pat)
{
    const unsigned w0 = (w & ((1u << S.cb_bits) - 1u)) | ((w >> 27) << UNIT_FLAG_SHIFT) | pat.z;   // (pat.z: the pattern's window shift, already at UNIT_SHIFT_SHIFT)
    return make_uint4(w0, pat.x, w0, pat.y);
}
// DERIVED units (hip_plan.h UNIT_DERIVED_CODE, plan_tile_ops.h): lanes 0-14 of the strip use the x the previous unit used one lane up (DPP row rotation: a DPP row is one strip),
// lane 15 the value it loaded itself; every other unit uses what it gathered.  `prev` = the x the previous unit of this strip used.
template <class X>
__device__ __forceinline__ X unit_x_use(X gathered, X prev, unsigned w0, int r)
{
    const X up = dpp_mov<0x12F>(prev);   // row_ror:15 = lane i reads lane i + 1
    return ((w0 >> UNIT_SHIFT_SHIFT) == UNIT_DERIVED_CODE && r != 15) ? up : gathered;
}
// first column of a classic unit's window of x: column block * 16, moved by the signed shift of a unit that took list entries (hip_plan.h UNIT_SHIFT_SHIFT)
__device__ __forceinline__ long long unit_x_base(unsigned w0) { return (long long)(w0 & 0xFFFFFFu) * 16 + ((int)w0 >> UNIT_SHIFT_SHIFT); }

// Descriptor word layout in LDS (16 B per unit, two identical-purpose halves so that a lane reads 8 B; HBM holds the
// 12-B form without the duplicate word, UDesc):
//   word 0 / word 2 : column block (24 bits) | flags << 24   (flag bit 0 = end of tile-row, bits 1-3 = row in strip,
//                     bit 4 = row unit, bits 5-7 = signed shift of the unit's window of x: a unit that took list entries, plan_tile_ops.h)
//   word 1          : column nibbles of rows 0-7  (row 0 in the top nibble)   [row unit: target row]
//   word 3          : column nibbles of rows 8-15
// Descriptors reach the lanes through LDS: one coalesced 16-B-per-lane load brings the descriptors of 16
// consecutive units (lane j loads unit j's), the strip parks them in LDS and every unit then costs one
// ds_read_b64 instead of one global load.  (Measured: per-unit descriptor loads, 6 % of the bytes, cost 19 %
// of the kernel — the CU's vector-memory pipeline is the bottleneck, not HBM)
constexpr int DCHUNK = 16;  // units per descriptor chunk
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
#ifndef NT_Y
#define NT_Y 1  // y is written once and not re-read by this kernel: streaming (nontemporal) stores keep it from displacing x in L2 (+1-2 %; per plan: DevStream::y_streaming)
#endif
#ifndef WCOO_HEAVY_CT
#define WCOO_HEAVY_CT 6  // sub-chunks (of 64 / 256 entries) per trip of the wavefront / workgroup entry phase
#endif
#ifndef MV_MIN_WAVES
#define MV_MIN_WAVES 6  // multi-vector kernel: 80 VGPRs (5 waves: 84 VGPRs, nvec 8 0.79 ms; 6: 0.74 ms; 7 spills: 1.01 ms)
#endif
#ifndef ECOO2_MIN_WAVES
#define ECOO2_MIN_WAVES 6  // workgroup entry mode: 80 VGPRs
#endif
#ifndef TILESPMV_UB
#define TILESPMV_UB 4   // units per batch of the unit loop (diagnostic builds: 8 with UNITS_MIN_WAVES=6 measured below)
#endif
#ifndef UNITS_MIN_WAVES
#define UNITS_MIN_WAVES 8  // waves per SIMD asked of the register allocator (64 VGPRs)
#endif
#ifndef POOL_ECOO2_MIN_WAVES
#define POOL_ECOO2_MIN_WAVES 5   // pooled plans, workgroup entry mode: 96 VGPRs (at 6 waves = 80 VGPRs the kernel spills 12 bytes)
#endif
#ifndef POOL_MIN_WAVES
#define POOL_MIN_WAVES 7   // pooled plans, per-strip entries: 14.5 KB of LDS per workgroup; 72 VGPRs (at 8 waves = 64 VGPRs the kernel spills 20 bytes and runs slower)
#endif

// ---- packed entry records (hip_plan.h ERec): value + (column - chunk base) << dest_bits | destination
__device__ __forceinline__ val_t erec_val(const ERec &r)
{
#if defined(TILESPMV_F32)
    return __uint_as_float(r.v);
#else
    return __hiloint2double((int)r.hi, (int)r.lo);
#endif
}

// ---- wave-cooperative entry phase of k_units<.., 1>: the COO entry lists of the wavefront's four strips, merged and ordered
// by column at plan time, are walked by all 64 lanes; products go to the owning strip's slab of the wavefront's part of
// s_y with ds_add (destination = strip-in-wavefront << 7 | row byte).  A wavefront's time follows its TOTAL entry
// count, not its longest strip; every load is a full 64-lane access; neighbouring lanes of a gather read the same or
// adjacent x lines, and the four wavefronts of a workgroup sweep the columns side by side, so they find each other's lines
// in the CU's L1.  Only this wavefront adds into its slabs: the order of the additions is fixed by the plan (bit-
// reproducible).  CT x 64 entries per trip: every record load of a trip (one 12-/8-byte lane load each; the chunk's column
// base comes through the scalar cache), then its gathers, then the adds.
template <int CT>
__device__ __forceinline__ void wave_entry_trips(const DevStream &S, const val_t *__restrict__ x, lacc_t *swave, int lane, int gb, int ge, int chunk0, int cfirst)
{
    const int db = S.dest_bits;
    const unsigned dmask = (1u << db) - 1u;
    const int clast = chunk0 + ((ge - 1 - gb) >> 6);
    for (int e0 = gb + 64 * cfirst; e0 < ge; e0 += 64 * CT) {
        ERec rr[CT]; unsigned cb[CT]; val_t xx[CT];
#pragma unroll
        for (int q = 0; q < CT; q++) {
            rr[q] = S.grec[min(e0 + 64 * q + lane, ge - 1)];
            cb[q] = S.gbase[__builtin_amdgcn_readfirstlane(min(chunk0 + ((e0 - gb) >> 6) + q, clast))];
        }
#pragma unroll
        for (int q = 0; q < CT; q++) xx[q] = x[(size_t)(cb[q] + (rr[q].w >> db))];
#pragma unroll
        for (int q = 0; q < CT; q++)
            if (e0 + 64 * q + lane < ge) atomicAdd(&swave[rr[q].w & dmask], (lacc_t)(erec_val(rr[q]) * xx[q]));
    }
}

// ---- workgroup-cooperative entry phase of k_units<.., 2> (and of the fallback kernel): the entries of the workgroup's 16 or
// 32 strips, merged and ordered by column at plan time, walked by all NT lanes.  Neighbouring lanes of a gather then read
// the same or adjacent x lines: on power-law matrices the number of distinct x lines per batch drops from 0.48 per entry (one
// strip at a time) to 0.15 (64 tile-rows at a time), and the CU's L1 -> L2 request rate is what bounds those matrices.
#ifndef WG_TRIP_PIPE
#define WG_TRIP_PIPE 0   // 1: the next trip's records are requested behind the current trip's gathers.  Measured (profiles/r03_entry_ablations.txt): power-law 8 M 0.1039 -> 0.1065 ms, KKT fp64 0.427 -> 0.435, webbase 13.1 -> 12.9 us at 4 x 256 per trip; 6 x 256 spills.  Off.
#endif
// NTL: the records are read with nontemporal loads (plans whose streams do not fit the Infinity Cache: the once-read stream
// then does not displace x in the L2s; DevStream::nt_stream).
__device__ __forceinline__ val_t gather_x(const val_t *p) { return TSPMV_DIAG_GATHER_X(p); }   // (diagnostic builds probe other cache policies here: hip_kernels_diag.h)

template <int CT, int NT, bool NTL>
__device__ __forceinline__ void wg_entry_trips(const ERec *__restrict__ rec, const unsigned *__restrict__ base, int chunk0, int db, bool ordered,
                                               const val_t *__restrict__ x, lacc_t *sy, int tid, int gb, int ge, int gs = -1)
{
    // [gs, ge) = the records to execute; gb = the list's begin, which chunk numbers count from (column panels execute a run that starts inside the list, even inside a chunk)
    if (gs < 0) gs = gb;
    const int e_first = gb + ((gs - gb) & ~63);
    const unsigned dmask = (1u << db) - 1u;
    const int wave = tid >> 6;
    const int clast = chunk0 + ((ge - 1 - gb) >> 6);
    TSPMV_DIAG_TRIP_DECL
    ERec rr[CT]; unsigned cb[CT];
    auto load_trip = [&](int e0, ERec (&r)[CT], unsigned (&c)[CT]) {   // unconditional, clamped: exact vmcnt
#pragma unroll
        for (int q = 0; q < CT; q++) {
            if constexpr (NTL) {   // (the adjacent nontemporal dword loads become one global_load_dwordx3 / dwordx2 nt)
                const unsigned *pw = reinterpret_cast<const unsigned *>(&rec[min(e0 + NT * q + tid, ge - 1)]);
                unsigned *rw = reinterpret_cast<unsigned *>(&r[q]);
#pragma unroll
                for (int z = 0; z < (int)(sizeof(ERec) / 4); z++) rw[z] = __builtin_nontemporal_load(pw + z);
            } else r[q] = rec[min(e0 + NT * q + tid, ge - 1)];
            c[q] = base[__builtin_amdgcn_readfirstlane(min(chunk0 + ((e0 - gb) >> 6) + (NT / 64) * q + wave, clast))];   // a wavefront's 64 records are one chunk
            TSPMV_DIAG_TRIP_RECORD(r, q, e0)
        }
    };
    if (e_first < ge) load_trip(e_first, rr, cb);
    for (int e0 = e_first; e0 < ge; e0 += NT * CT) {
        val_t xx[CT];
        if (!WG_TRIP_PIPE && e0 > e_first) load_trip(e0, rr, cb);
#pragma unroll
        for (int q = 0; q < CT; q++) xx[q] = gather_x(&x[(size_t)(cb[q] + (rr[q].w >> db))]);
        TSPMV_DIAG_TRIP_GATHERS
        // the next trip's records go in flight behind this trip's gathers (loads return in issue order: the gathers are waited
        // for with the prefetch still outstanding); the last trip re-requests its own (clamped) records, which costs nothing
        ERec rn[CT]; unsigned cn[CT];
        if (WG_TRIP_PIPE) load_trip(min(e0 + NT * CT, gb + (ge - 1 - gb) / (NT * CT) * (NT * CT)), rn, cn);
        if constexpr (TSPMV_DIAG_TRIP_ADDS_REPLACED) { TSPMV_DIAG_TRIP_ADDS }
        else if (ordered) {
            // the wavefronts add in turn: the order of the additions into one y element is then fixed by the plan (entry
            // order inside a wavefront instruction, instruction order inside a wavefront, wavefront 0..NT/64-1 inside a trip), not
            // by timing, and two launches give the same bits (the reference's atomicAdd, src/tilespmv_cuda.h:784-790, does not)
            for (int w = 0; w < NT / 64; w++) {
                if (wave == w) {
#pragma unroll
                    for (int q = 0; q < CT; q++)
                        if (e0 + NT * q + tid < ge && e0 + NT * q + tid >= gs) atomicAdd(&sy[rr[q].w & dmask], (lacc_t)(erec_val(rr[q]) * xx[q]));
                }
                __syncthreads();
            }
        } else {
#pragma unroll
            for (int q = 0; q < CT; q++)
                if (e0 + NT * q + tid < ge && e0 + NT * q + tid >= gs) atomicAdd(&sy[rr[q].w & dmask], (lacc_t)(erec_val(rr[q]) * xx[q]));
        }
        if (WG_TRIP_PIPE) {
#pragma unroll
            for (int q = 0; q < CT; q++) { rr[q] = rn[q]; cb[q] = cn[q]; }
        }
    }
}

// ================================================================================================
// Very-sparse fallback: y[row] += sum_j val[j] * x[col[j]] over the extracted matrix (the reference hands it to CSR5,
// src/tilespmv_cuda.h:1011-1029,:1080; kernels src/external/CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:277-420).
// nnz-balanced like CSR5 — one workgroup per row block of <= FB_CAP nonzeros (and <= FB_ROWS rows), a longer row cut into
// pieces that add atomically as CSR5's calibrate step does (:315-384) — but with what bounds this chip on such matrices in
// mind: the block's nonzeros are ordered by column at plan time and walked by all 256 lanes, so that the
// lanes of one gather share x lines, and the row sums are accumulated in LDS.
// ================================================================================================
__global__ __launch_bounds__(256) void k_fallback_entries(DevPlan P, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    __shared__ lacc_t s_acc[FB_ROWS];
    const int tid = threadIdx.x;
    const int4 b = P.f_blk[blockIdx.x];
    const int nrows = b.y < 0 ? 1 : b.y;
    for (int i = tid; i < nrows; i += 256) s_acc[i] = 0;
    __syncthreads();
    wg_entry_trips<6, 256, false>(P.f_rec, P.f_base, b.z >> 6, FB_DEST_BITS, P.f_ordered != 0, x, s_acc, tid, b.z, b.w);   // a block's list starts on a chunk boundary
    __syncthreads();
    if (b.y < 0) {
        if (tid == 0) atomicAdd(&y[(long long)P.f_row0 + b.x], (val_t)s_acc[0]);
    } else {
        // every row of the block is updated, also those whose sum is zero: the kernel's time must not depend on the values (round 2 skipped
        // exact zeros, which made an all-zero x look 20-30 % faster to the autotuner)
        for (int i = tid; i < nrows; i += 256) y[(long long)P.f_row0 + b.x + i] = (val_t)((lacc_t)y[(long long)P.f_row0 + b.x + i] + s_acc[i]);
    }
}

hipError_t launch_fallback(const DevPlan &P, const val_t *x, val_t *y, hipStream_t st)
{
    if (P.f_nblk > 0) hipLaunchKernelGGL(k_fallback_entries, dim3((unsigned)P.f_nblk), dim3(256), 0, st, P, x, y);
    return hipGetLastError();
}

// ECOO: how the COO entry lists are executed — 0 per 16-lane strip (regular matrices: a handful of entries per strip),
// 1 per wavefront (the four strips' lists concatenated), 2 per workgroup (merged + column-ordered list, see above).
// GPB: strips (16-lane groups) per workgroup — 16 (256 threads) or, for the workgroup entry mode on large entry-heavy shards, 32
// (512 threads: twice as many tile-rows share one column-ordered list, so fewer distinct x lines per entry; same waves per SIMD).
// (Retired in round 6, both measured slower than what replaced them: x windows staged in LDS — DESIGN S6.9 — and the slab-paced entry phase — S6.17.  The XCD remap is a run-time
// scalar branch now (xcd_chunk > 0) instead of a template axis.)
extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
// CD: dictionary plans (4-B descriptors, above).  The descriptor words are loaded two chunks ahead, the pattern of a chunk is
// gathered from the dictionary one chunk ahead (when its word has arrived), so neither hop is waited for in the unit loop.
// NTS: value and entry-record loads are nontemporal (plans larger than the Infinity Cache, DevStream::nt_stream).
// POOL: pooled plans (hip_plan.h "pooled units", round 5): a unit is up to 16 nonzeros of a tile-row inside one 16-column window of x — slot s = value, column-offset nibble, row nibble —
// so a lane no longer owns a row: it gathers x[base + its column nibble] and adds its product to the strip's slab of s_y with ds_add (destination = tile-row in strip, its row
// nibble); there is no register accumulator, no end-of-row handling and no "rows without units": the slab is zeroed up front, entries and units add into it, y is stored from it.
// The row nibbles travel like the descriptors (8 bytes per unit, one coalesced lane load per chunk of 16 units, parked in LDS: + 2 KB per workgroup -> 7 workgroups per CU).
// WIDE (with POOL; hip_plan.h "wide pooled units", csr_form 3): windows of 256 columns — a slot's column offset is a byte (16 bytes per unit in S.ucol, parked in s_c), the descriptor's nibble words hold the ROW nibbles.
template <int UB, int ECOO, int GPB, bool CD, bool NTS, bool POOL = false, bool WIDE = false>
__global__ __launch_bounds__(16 * GPB, ECOO == 1 ? 4 : ECOO == 2 ? (POOL ? POOL_ECOO2_MIN_WAVES : ECOO2_MIN_WAVES) : POOL ? (WIDE ? 6 : POOL_MIN_WAVES) : UNITS_MIN_WAVES) void k_units(DevStream S, int rowA, int colA, int xcd_chunk, val_t *__restrict__ partial,
                                               const val_t *__restrict__ x, val_t *__restrict__ y)
{
    static_assert(DCHUNK % UB == 0 && UB % UNIT_GROUP == 0, "a batch never straddles a descriptor chunk and is whole value groups");
    static_assert(GPB == 16 || (GPB == 32 && ECOO == 2) || (GPB == 8 && ECOO != 2 && !NTS), "512-thread workgroups exist for the workgroup entry mode only, 128-thread ones for small grids without it");
    static_assert(!(NTS && ECOO == 1), "nontemporal streams: large plans only (entry mode 1 = small grids)");
    static_assert(!POOL || GPB == 16 || GPB == 8, "pooled plans: 256-thread workgroups (128 on small grids)");
    static_assert(!WIDE || (POOL && !CD), "wide windows: pooled plans, 12-B descriptors + 16 B of column offsets");
    constexpr int GROUPS_PER_BLOCK = GPB;
    constexpr int SROWS = POOL ? POOL_STRIP_ROWS : STRIP_MAX_ROWS;   // tile-rows per strip the LDS slabs are sized for
    constexpr bool NT = NTS;  // nontemporal value loads
#ifndef TILESPMV_NT_DESC
#define TILESPMV_NT_DESC 0
#endif
#ifndef TILESPMV_NT_COO0
#define TILESPMV_NT_COO0 0
#endif
    constexpr bool NT_DESC = NTS && TILESPMV_NT_DESC, NT_COO0 = NTS && TILESPMV_NT_COO0;
    // Pooled plans of the fp32 build keep a SECOND copy of the slabs (PCOPY): the slots of a unit are in row order, so the nonzeros of one row sit in neighbouring lanes, and lanes of
    // odd / even slot add into different copies — two lanes of one LDS atomic then (almost) never hit one address.  A same-address pair costs the fp32 build a quarter of its time
    // (fem3_68: 0.119 ms, 0.088 with lane-private addresses: profiles/r05_pool_ablations.txt; the fp64 build is bound by its bytes and gains nothing).  The copy sits 16 doubles
    // off a multiple of the bank count, so that the two halves of a pair also fall into different banks; the copies are summed when the strip is done.
    constexpr bool PCOPY = POOL && sizeof(val_t) == 4;
    constexpr int SLAB = GROUPS_PER_BLOCK * SROWS * 16;
    __shared__ lacc_t s_yall[SLAB + (PCOPY ? SLAB + 16 : 0)];   // (lacc_t: fp64 in both builds, see its typedef)
    lacc_t (*s_y)[SROWS][16] = reinterpret_cast<lacc_t (*)[SROWS][16]>(&s_yall[0]);
    lacc_t *s_y1 = &s_yall[PCOPY ? SLAB + 16 : 0];
    __shared__ uint4 s_d[GROUPS_PER_BLOCK][DCHUNK];
    __shared__ uint2 s_r[POOL && !WIDE ? GROUPS_PER_BLOCK : 1][POOL && !WIDE ? DCHUNK : 1];   // pooled plans: row nibbles of the parked descriptor chunk
    __shared__ uint4 s_c[WIDE ? GROUPS_PER_BLOCK : 1][WIDE ? DCHUNK : 1];                   // wide pooled plans: column-offset bytes of the parked chunk
    TSPMV_DIAG_UNITS_LDS_PAD
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group), each
    // with a private L2; xcd_chunk > 0 gives every XCD runs of xcd_chunk consecutive workgroups inside
    // windows of 8 x xcd_chunk (bijective for any grid size, cdna_hip_programming.md T1).  Speed only.
    unsigned bid = blockIdx.x;
    if (xcd_chunk > 0) {   // (kernel argument: a scalar branch)
        const unsigned C = (unsigned)xcd_chunk, W = 8u * C, win = bid / W, off = bid % W, k = off & 7u;
        if ((win + 1) * W <= gridDim.x) bid = win * W + k * C + (off >> 3);
    }
    TSPMV_STAMP_DECL
    TSPMV_STAMP(0);
    const long long task_id = (long long)bid * GROUPS_PER_BLOCK + g;
    const bool have = task_id < S.ntasks;
    constexpr bool WCOO = ECOO == 1;
    if (ECOO == 1) { if ((long long)bid * GROUPS_PER_BLOCK + (g & ~3) >= S.ntasks) return; }  // whole wavefronts leave together (wave-cooperative entry phase)
    else if (ECOO == 0 && !have) return;                                                      // ECOO == 2: every wavefront reaches the barriers
    int4 t0 = make_int4(0, 0, 0, 0), t1 = make_int4(0, -1, 0, 0);
    if (have) {
        t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
        t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    }
    const int unit_begin = t0.x, unit_end = t0.y, coo_begin = t0.z, coo_end = t0.w;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    const unsigned nounit = (unsigned)t1.z;
    const bool side = POOL || coo_end > coo_begin;   // the strip's slab of s_y holds sums (entries; in pooled plans everything)
    if constexpr (POOL) {   // the slab is zeroed before anything adds into it (the entry phases below sit behind a fence / barrier of their own)
        for (int k = 0; k < nrows; k++) { s_y[g][k][r] = 0; if constexpr (PCOPY) s_y1[(g * SROWS + k) * 16 + r] = 0; }
        wave_lds_fence();
    }
    // values are stored in groups of G = UNIT_GROUP units of one task (hip_plan.hip): row r of the group that starts at
    // task-relative unit j (a multiple of G) sits at uval[(unit_begin + j) * 16 + G r .. + G - 1]; a batch of UB units
    // is UB / G sixteen-byte loads per lane
    constexpr int G = UNIT_GROUP;
    typedef val_t grp_t __attribute__((ext_vector_type(UNIT_GROUP)));
    const grp_t *__restrict__ ugrp = reinterpret_cast<const grp_t *>(S.uval) + r;
    const int last = unit_end - 1;
    const int last_grp = unit_begin + (unit_end - 1 - unit_begin) / UNIT_GROUP * UNIT_GROUP;  // first unit of the task's last group
    const bool have_units = unit_begin < unit_end;
    const long long xlast = (long long)colA - 1;  // row units of a partial last column block: zero payload, clamped x index
    const int ncoo = coo_end - coo_begin;
    uint4 dcur = make_uint4(0u, 0u, 0u, 0u), dnext = dcur;
    uint2 rcur = make_uint2(0u, 0u), rnext = rcur;   // POOL: row nibbles of the chunks in dcur / dnext
    const uint2 *__restrict__ urw = reinterpret_cast<const uint2 *>(S.urow);
    uint4 ccur = make_uint4(0u, 0u, 0u, 0u), cnext = ccur;   // WIDE: column-offset bytes of the chunks in dcur / dnext
    unsigned wnn = 0;   // CD: descriptor word of the chunk after `dnext`
    const unsigned *__restrict__ udw = reinterpret_cast<const unsigned *>(S.udesc);
    // POOL + CD (pooled dictionary plans, round 5): a unit's descriptor in HBM is 8 bytes — word 0 (window base | tile-row in strip) and the id of its 16-byte pattern (the 16 column
    // nibbles and the 16 row nibbles) in S.pdict, which stays in the vector L1 / L2: natural-order meshes use a few dozen patterns (fem3_68: 54).  Same staging as the classic
    // dictionary: words two chunks ahead, the pattern gathered one chunk ahead, so neither hop is waited for in the unit loop.  Round 6: 4-byte words where everything fits one (pool_desc).
    uint2 wnn2 = make_uint2(0u, 0u);
    val_t v[UB];
    auto unit_prologue = [&]() {  // descriptor chunks 0 and 1, first value batch: in flight across the entry phase
        if (have_units) {
            if constexpr (CD && POOL) {   // pooled dictionary plans: 8-byte descriptors (word 0, pattern id)
                const uint2 a = pool_desc(S, min(unit_begin + r, last)), b = pool_desc(S, min(unit_begin + DCHUNK + r, last));
                dcur.x = a.x; dcur.y = a.y; dnext.x = b.x; dnext.y = b.y;
            } else if constexpr (CD) {
                dcur.x = stream_load<NT_DESC>(udw + min(unit_begin + r, last));
                dnext.x = stream_load<NT_DESC>(udw + min(unit_begin + DCHUNK + r, last));
            } else {
                dcur = load_udesc_raw(S.udesc, min(unit_begin + r, last));
                dnext = load_udesc_raw(S.udesc, min(unit_begin + DCHUNK + r, last));
                if constexpr (WIDE) { ccur = S.ucol[min(unit_begin + r, last)]; cnext = S.ucol[min(unit_begin + DCHUNK + r, last)]; }
                else if constexpr (POOL) { rcur = urw[min(unit_begin + r, last)]; rnext = urw[min(unit_begin + DCHUNK + r, last)]; }   // (12-byte descriptors + 8 bytes of row nibbles)
            }
#pragma unroll
            for (int k = 0; k < UB; k += G) {
                const grp_t pv = stream_load<NT>(ugrp + (long long)min(unit_begin + k, last_grp) * (16 / G));
#pragma unroll
                for (int q = 0; q < G; q++) v[k + q] = pv[q];
            }
        }
    };

    // descriptors (from the chunk parked in LDS) and x gathers of one unit batch; j0 = position of the batch in the chunk
    const uint2 *sd = reinterpret_cast<const uint2 *>(&s_d[g][0]) + (r >> 3);  // this lane's 8-B half of a descriptor
    uint2 d[UB];
    unsigned rw[UB];   // POOL: this lane's half of the unit's row nibbles
    val_t xv[UB];
    const unsigned *sr = reinterpret_cast<const unsigned *>(&s_r[POOL && !WIDE ? g : 0][0]) + (r >> 3);
    const unsigned char *sc = reinterpret_cast<const unsigned char *>(&s_c[WIDE ? g : 0][0]) + r;   // this lane's byte of a unit's 16 column offsets
    auto park_first = [&]() {   // chunk 0 into LDS (CD: the patterns of chunks 0 and 1 are gathered here, the word of chunk 2 loaded)
        if constexpr (CD && POOL) {
            const uint4 p0 = S.pdict[dcur.y], p1 = S.pdict[dnext.y];
            wnn2 = pool_desc(S, min(unit_begin + 2 * DCHUNK + r, last));
            s_d[g][r] = make_uint4(dcur.x, p0.x, dcur.x, p0.y); s_r[g][r] = make_uint2(p0.z, p0.w);
            dnext = make_uint4(dnext.x, p1.x, 0u, p1.y); rnext = make_uint2(p1.z, p1.w);
        } else if constexpr (CD) {
            const uint4 p0 = udict_of(S, dcur.x), p1 = udict_of(S, dnext.x);
            wnn = stream_load<NT_DESC>(udw + min(unit_begin + 2 * DCHUNK + r, last));
            dnext.y = p1.x; dnext.z = p1.z; dnext.w = p1.y;
            s_d[g][r] = udesc_expand(S, dcur.x, p0);
        } else s_d[g][r] = udesc_park_form(dcur);
        if constexpr (WIDE) s_c[g][r] = ccur;
        else if constexpr (POOL && !CD) s_r[g][r] = rcur;
    };
    auto fetch_batch = [&](int j0) {
#pragma unroll
        for (int k = 0; k < UB; k++) d[k] = sd[2 * (j0 + k)];
        if constexpr (WIDE) {   // the descriptor's nibble half = this lane's row nibbles; the column offset is its byte of the unit's 16
#pragma unroll
            for (int k = 0; k < UB; k++) rw[k] = d[k].y;
#pragma unroll
            for (int k = 0; k < UB; k++) xv[k] = x[min((long long)(d[k].x & POOL_BASE_MASK) + (long long)sc[16 * (j0 + k)], xlast)];
            return;
        }
        if constexpr (POOL) {
#pragma unroll
            for (int k = 0; k < UB; k++) rw[k] = sr[2 * (j0 + k)];
#pragma unroll
            for (int k = 0; k < UB; k++) xv[k] = TSPMV_DIAG_POOL_X(x[min((long long)(d[k].x & POOL_BASE_MASK) + (long long)((d[k].y >> (28 - 4 * (r & 7))) & 15u), xlast)], d[k]);
            return;
        }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const unsigned fl = d[k].x >> 24;
            const unsigned nib = (fl & UNIT_ROWUNIT) ? (unsigned)r : (d[k].y >> (28 - 4 * (r & 7))) & 15u;
            if (TSPMV_DIAG_UNIT_GATHER_SKIP(k)) xv[k] = xv[k - 1];
            else if ((d[k].x >> UNIT_SHIFT_SHIFT) == UNIT_DERIVED_CODE && r != 15) xv[k] = 0;   // derived unit: only lane 15 loads (unit_x_use gives the others the previous unit's x)
            else xv[k] = x[TSPMV_DIAG_UNIT_X(min(unit_x_base(d[k].x) + nib, xlast))];
        }
    };

    if constexpr (ECOO == 2) {
        const int4 wr = S.wg_coo[bid];
        TSPMV_STAMP_WAIT(1);   // task and list range have arrived
        // a strip with entries adds its slab at the end, so its slab is zeroed even when the workgroup's list is empty: in a column-panelled plan (x_panels > 1) all of a
        // strip's entries may sit in the other panels' lists
        if (wr.y > wr.x) {  // workgroup-uniform
            if (side) for (int k = 0; k < nrows; k++) s_y[g][k][r] = 0;
            __syncthreads();
        } else if ((S.panel_merge > 0 || S.slice_passes > 0) && side) {
            for (int k = 0; k < nrows; k++) s_y[g][k][r] = 0;
            wave_lds_fence();
        }
        // (issuing the unit prologue after the entry phase instead frees 16 VGPRs — 8 waves per SIMD at 6 x 256 per trip, or 8 x 256 at
        // 6 waves — and changes nothing: power-law 8 M 0.1026-0.1034 ms either way, profiles/r03_entry_ablations.txt: bytes in flight are not the limit)
        // pipelined trips on top of that (78 VGPRs at 6 x 256, no spill): 0.1050-0.1062 against 0.1031-0.1046 — slightly worse
        unit_prologue();
        TSPMV_STAMP_WAIT(2);   // prologue has arrived (the entry loads are inside the trips)
        if (wr.y > wr.x) {
            int ge = wr.y;   // column-panelled launch: this kernel takes the first panel_merge panels of the list, k_entries_acc the rest
            if (GPB == 16 && S.panel_merge > 0) ge = S.panel_off[(size_t)bid * (size_t)(S.x_panels + 1) + (size_t)min(S.x_panels, S.panel_merge)];
            if (GPB == 16 && S.slice_passes > 0) ge = wr.x;   // column slices pinned to XCDs: the whole list belongs to k_entries_xcd
            wg_entry_trips<WCOO_HEAVY_CT, 16 * GPB, NTS>(S.grec, S.gbase, wr.z, S.dest_bits, S.coo_ordered != 0, x, &s_y[0][0][0], tid, wr.x, ge);
            __syncthreads();
        }
        TSPMV_STAMP_WAIT(3);   // entry phase done
    } else if constexpr (WCOO) {
        // ---- small grids (entry mode 1 is chosen when the whole grid is resident at once): the kernel is a chain of
        // round trips, so everything that can be in flight together is: task -> {unit prologue, entry loads} -> {x gathers
        // of the first unit batch, x gathers of the entries} -> adds -> unit loop.  Registers are not a constraint here
        // (4 waves/SIMD asked of the allocator).
        constexpr int CT = 6;
        const int lane = tid & 63;
        const int4 wr = S.wg_coo[(long long)bid * (GROUPS_PER_BLOCK / 4) + (g >> 2)];  // this wavefront's merged list
        const int tot = wr.y - wr.x;
        lacc_t *swave = &s_y[g & ~3][0][0];  // the wavefront's four slabs of STRIP_MAX_ROWS x 16 values
        TSPMV_STAMP_WAIT(1);   // task and list range have arrived
        unit_prologue();
        ERec rr[CT]; unsigned cbase[CT]; val_t xx[CT];
        const int db = S.dest_bits;
        if (tot > 0) {
            if (side) for (int k = 0; k < nrows; k++) s_y[g][k][r] = 0;
            const int clast = wr.z + ((tot - 1) >> 6);
#pragma unroll
            for (int q = 0; q < CT; q++) {
                rr[q] = S.grec[min(wr.x + 64 * q + lane, wr.y - 1)];
                cbase[q] = S.gbase[__builtin_amdgcn_readfirstlane(min(wr.z + q, clast))];
            }
        }
        if (have_units) {  // waits for the descriptor chunk only (older than the entry loads)
            park_first();
            wave_lds_fence();
            fetch_batch(0);
        } else if (tot > 0) wave_lds_fence();
        TSPMV_STAMP_WAIT(2);   // prologue + entry loads (and the first unit batch's gathers) have arrived
        if (tot > 0) {
            const unsigned dmask = (1u << db) - 1u;
#pragma unroll
            for (int q = 0; q < CT; q++) xx[q] = x[(size_t)(cbase[q] + (rr[q].w >> db))];
#pragma unroll
            for (int q = 0; q < CT; q++)
                if (wr.x + 64 * q + lane < wr.y) atomicAdd(&swave[rr[q].w & dmask], (lacc_t)(erec_val(rr[q]) * xx[q]));
            if (tot > 64 * CT) wave_entry_trips<CT>(S, x, swave, lane, wr.x, wr.y, wr.z, CT);
            wave_lds_fence();
        }
        TSPMV_STAMP_WAIT(3);   // entry phase done
    } else {
    // ---- issue order: first COO chunk, descriptor chunk 0 (+1), first value batch: all in flight together.
    // Strips with many COO entries (> coo_heavy_min, default 32: irregular matrices) run their entry list first,
    // 6 x 16 entries per trip with every load of a trip in flight before its gathers, and only then start the
    // unit pipeline; the others keep the unit prologue in flight across their (short) entry list.
    constexpr int CT = 6;  // sub-chunks of 16 entries per trip
    const bool coo_heavy = ncoo > S.coo_heavy_min;
    if (side) {
        for (int k = 0; k < nrows; k++) s_y[g][k][r] = 0;
        wave_lds_fence();
    }
    if (coo_heavy) {
        for (int e0 = coo_begin; e0 < coo_end; e0 += 16 * CT) {
            unsigned rb[CT]; int cc[CT]; val_t cv[CT], xx[CT];
#pragma unroll
            for (int q = 0; q < CT; q++) {
                const int e = min(e0 + 16 * q + r, coo_end - 1);
                rb[q] = S.crow[e]; cc[q] = S.ccol[e]; cv[q] = S.cval[e];
            }
#pragma unroll
            for (int q = 0; q < CT; q++) xx[q] = x[cc[q]];
#pragma unroll
            for (int q = 0; q < CT; q++)
                if (e0 + 16 * q + r < coo_end) atomicAdd(&s_y[g][rb[q] >> 4][rb[q] & 15u], (lacc_t)(cv[q] * xx[q]));
        }
        wave_lds_fence();
    }
    unsigned rb0 = 0; int cc0 = 0; val_t cv0 = 0;
    const bool coo0 = TSPMV_DIAG_COO0_ON && side && !coo_heavy && (coo_begin + r < coo_end);
    if (coo0) { rb0 = stream_load<NT_COO0>(S.crow + coo_begin + r); cc0 = stream_load<NT_COO0>(S.ccol + coo_begin + r); cv0 = stream_load<NT_COO0>(S.cval + coo_begin + r); }
    unit_prologue();
    if (side && !coo_heavy) {  // up to coo_heavy_min entries: 16 with the prologue loads, the rest 4 x 16 per trip
        if (coo0) atomicAdd(&s_y[g][rb0 >> 4][rb0 & 15u], (lacc_t)(cv0 * x[TSPMV_DIAG_COO0_X(cc0)]));
        for (int e0 = coo_begin + 16; e0 < coo_end; e0 += 64) {
            unsigned rb[4]; int cc[4]; val_t cv[4], xx[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = min(e0 + 16 * q + r, coo_end - 1);
                rb[q] = S.crow[e]; cc[q] = S.ccol[e]; cv[q] = S.cval[e];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) xx[q] = x[cc[q]];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (e0 + 16 * q + r < coo_end) atomicAdd(&s_y[g][rb[q] >> 4][rb[q] & 15u], (lacc_t)(cv[q] * xx[q]));
        }
        wave_lds_fence();
    }
    }

    val_t acc = 0;
    // A finished tile-row parks its 16 results in s_y; y is written once per strip at the end with 16-B
    // lane stores.  (Stores share the in-order vmcnt queue with the loads on CDNA4: a store in the middle
    // of the unit loop makes every later counted wait also wait for its write acknowledge.)
    auto retire = [&](val_t prod, unsigned flags, unsigned word1) {
        if constexpr (POOL) {   // flags = word 0 >> 24 (tile-row in strip in its top nibble), word1 = this lane's half of the row nibbles
            const unsigned dest = ((flags >> (POOL_KR_SHIFT - 24)) & 7u) * 16u + ((word1 >> (28 - 4 * (r & 7))) & 15u);
            if (TSPMV_DIAG_POOL_ADD(dest, prod)) return;
            if constexpr (PCOPY) atomicAdd(((r & 1) ? s_y1 : &s_y[0][0][0]) + g * (SROWS * 16) + dest, (lacc_t)prod);
            else atomicAdd(&s_y[g][0][0] + dest, (lacc_t)prod);
            return;
        }
        if (flags & UNIT_ROWUNIT) {  // dense-row unit: lanes hold one row's products
            prod = strip_allreduce(prod);
            if (r != (int)(word1 & 15u)) prod = 0;
        }
        acc += prod;
        if (flags & UNIT_EOR) {
            const int kr = (int)((flags >> UNIT_ROW_SHIFT) & 7u);
            if constexpr (sizeof(val_t) == sizeof(lacc_t)) {
                lacc_t out = acc;
                if (side) out += s_y[g][kr][r];
                s_y[g][kr][r] = out;
            } else {
                // fp32 build: the row's entry sums (fp64, complete: the entry phase is over) are read and the 16 FINAL results go back as floats into the head of the
                // row's 128 bytes — lane r's 4 bytes overlap the doubles of lanes r/2, which every lane of the strip has read one instruction earlier — so that
                // the y store at the end reads 16 bytes per lane as in the fp64 build instead of narrowing four doubles
                val_t out = acc;
                if (side) out = (val_t)((lacc_t)acc + s_y[g][kr][r]);
                reinterpret_cast<val_t *>(&s_y[g][kr][0])[r] = out;
            }
            acc = 0;
        }
    };
    if (have_units) {  // phase 2: units, value loads software-pipelined by one batch
        if (ECOO != 1) {  // (entry mode 1 parked the first chunk and fetched the first batch before its entry phase)
            park_first();
            wave_lds_fence();
        }
        int chunk_end = unit_begin + DCHUNK;  // first unit NOT described by the chunk in LDS
        val_t xprev = 0;   // classic plans: the x the previous unit used (derived units, unit_x_use)
        for (int u = unit_begin; u < unit_end; u += UB) {
            if (u == chunk_end) {  // next descriptor chunk: already in registers, fetch the one after it
                wave_lds_fence();
                if constexpr (CD && POOL) { s_d[g][r] = make_uint4(dnext.x, dnext.y, dnext.x, dnext.w); s_r[g][r] = rnext; }
                else if constexpr (CD) s_d[g][r] = udesc_expand(S, dnext.x, make_uint4(dnext.y, dnext.w, dnext.z, 0u));
                else s_d[g][r] = udesc_park_form(dnext);
                if constexpr (WIDE) { s_c[g][r] = cnext; cnext = S.ucol[min(chunk_end + DCHUNK + r, last)]; }
                else if constexpr (POOL && !CD) { s_r[g][r] = rnext; rnext = urw[min(chunk_end + DCHUNK + r, last)]; }
                wave_lds_fence();
                chunk_end += DCHUNK;
                if constexpr (CD && POOL) {
                    const uint4 p = S.pdict[wnn2.y];   // (its words were loaded a chunk ago)
                    dnext = make_uint4(wnn2.x, p.x, 0u, p.y); rnext = make_uint2(p.z, p.w);
                    wnn2 = pool_desc(S, min(chunk_end + DCHUNK + r, last));
                } else if constexpr (CD) {
                    const uint4 p = udict_of(S, wnn);   // (its word was loaded a chunk ago)
                    dnext = make_uint4(wnn, p.x, p.z, p.y);
                    wnn = stream_load<NT_DESC>(udw + min(chunk_end + DCHUNK + r, last));
                } else dnext = load_udesc_raw(S.udesc, min(chunk_end + r, last));
            }
            if (!(ECOO == 1 && u == unit_begin)) fetch_batch(u - (chunk_end - DCHUNK));
            val_t vn[UB];
#pragma unroll
            for (int k = 0; k < UB; k += G) {  // unconditional (clamped to the task's last group): exact vmcnt
                const grp_t pv = stream_load<NT>(ugrp + (long long)min(u + UB + k, last_grp) * (16 / G));
#pragma unroll
                for (int q = 0; q < G; q++) vn[k + q] = pv[q];
            }
#pragma unroll
            for (int k = 0; k < UB; k++)
            {
                if constexpr (POOL) {
                    // unconditional adds (a unit past the task's end adds 0 to a row of this strip's slab: its descriptor is the clamped load of the task's last unit): with the add
                    // under a branch the compiler sinks the unit's gather into the branch and waits for it with vmcnt(0) — every unit then pays a full memory round trip
                    retire((u + k < unit_end) ? v[k] * xv[k] : (val_t)0, d[k].x >> 24, rw[k]);
                } else {
                    const val_t xu = unit_x_use(xv[k], xprev, d[k].x, r);
                    xprev = xu;
                    if (u + k < unit_end) retire(v[k] * xu, d[k].x >> 24, d[k].y);
                }
            }
#pragma unroll
            for (int k = 0; k < UB; k++) v[k] = vn[k];
        }
    }
    TSPMV_STAMP_WAIT(4);       // unit loop done
    if constexpr (POOL) {   // every add of this wavefront into the slab is behind us
        wave_lds_fence();
        if constexpr (PCOPY) {
            for (int k = 0; k < nrows; k++) s_y[g][k][r] += s_y1[(g * SROWS + k) * 16 + r];
            wave_lds_fence();
        }
    }
    if (part >= 0) {
        val_t out = acc;
        if (side) out = (val_t)((lacc_t)acc + s_y[g][0][r]);
        if (S.ifix_count == nullptr || nounit == 0xFFFFFFFFu) {
            partial[(long long)part * 16 + r] = out;  // k_fixup_split adds the slots up after all passes
        } else {
            // All pieces of this tile-row run in this kernel: the piece that finishes last adds the slots up, in slot
            // order (same sum as k_fixup_split).  Slots and counter are agent-scope atomics (performed at the device's
            // point of coherence, past the per-XCD L2s), the counter is bumped only after this strip's 16 slot stores
            // have been acknowledged, and the slot loads are issued only after the counter value has come back.
            __hip_atomic_store(&partial[(long long)part * 16 + r], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const FixRow f = S.ifix[nounit];
            unsigned prev = 0;
            if (r == 0) prev = __hip_atomic_fetch_add(&S.ifix_count[nounit], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = (unsigned)__shfl((int)prev, tid & 48, 64);  // lane 0 of this strip
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (prev == (unsigned)f.count - 1u) {
                val_t sum = 0;
                for (int k0 = 0; k0 < f.count; k0 += 8) {  // 8 slot loads in flight, added in slot order
                    val_t sv[8];
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        sv[j] = __hip_atomic_load(&partial[(long long)(f.first + min(k0 + j, f.count - 1)) * 16 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int j = 0; j < 8; j++) if (k0 + j < f.count) sum += sv[j];
                }
                const long long yi = (long long)f.row * 16 + r;
                if (yi < rowA) y[yi] = sum;
                if (r == 0) __hip_atomic_store(&S.ifix_count[nounit], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            }
        }
    } else {
        if constexpr (POOL) {
            if constexpr (sizeof(val_t) != sizeof(lacc_t)) {   // fp32 build: the fp64 sums go back as floats into the head of each row's 128 bytes (the form the store below reads; lane r's 4 bytes
                for (int k = 0; k < nrows; k++) {              // overlap the doubles of lanes r / 2, which every lane of the strip has read one instruction earlier)
                    const val_t o = (val_t)s_y[g][k][r];
                    reinterpret_cast<val_t *>(&s_y[g][k][0])[r] = o;
                }
            }
        } else if constexpr (sizeof(val_t) == sizeof(lacc_t)) {
            if (!side) {  // rows without any unit and no COO contribution are zero
                unsigned m = nounit;
                while (m) { const int kr = __ffs((int)m) - 1; m &= m - 1; s_y[g][kr][r] = 0; }
            }
        } else {          // fp32 build: rows without units hold fp64 entry sums (or nothing): into the float form of the retired rows
            unsigned m = nounit;
            while (m) {
                const int kr = __ffs((int)m) - 1; m &= m - 1;
                const val_t o = side ? (val_t)s_y[g][kr][r] : (val_t)0;
                reinterpret_cast<val_t *>(&s_y[g][kr][0])[r] = o;
            }
        }
        wave_lds_fence();
        constexpr int VEC = 16 / (int)sizeof(val_t);  // values per 16-B lane store
        const lacc_t *res = &s_y[g][0][0];
        const long long ybase = (long long)row0 * 16;
        for (int i = r * VEC; i < 16 * nrows; i += 16 * VEC) {
            if constexpr (sizeof(val_t) == sizeof(lacc_t)) {   // fp64: the 16 bytes go out as they sit in LDS (one ds_read_b128, one store)
                if (ybase + i + VEC <= rowA) {
                    if (NT_Y && S.y_streaming) __builtin_nontemporal_store(*reinterpret_cast<const v4u_t *>(res + i), reinterpret_cast<v4u_t *>(y + ybase + i));
                    else *reinterpret_cast<uint4 *>(y + ybase + i) = *reinterpret_cast<const uint4 *>(res + i);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; q++) if (ybase + i + q < rowA) y[ybase + i + q] = (val_t)res[i + q];
                }
            } else {                                           // fp32: the rows hold 16 floats each at the head of their 128 bytes (retire / flush above)
                const val_t *rf = reinterpret_cast<const val_t *>(res) + (i >> 4) * 32 + (i & 15);
                if (ybase + i + VEC <= rowA) {
                    if (NT_Y && S.y_streaming) __builtin_nontemporal_store(*reinterpret_cast<const v4u_t *>(rf), reinterpret_cast<v4u_t *>(y + ybase + i));
                    else *reinterpret_cast<uint4 *>(y + ybase + i) = *reinterpret_cast<const uint4 *>(rf);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; q++) if (ybase + i + q < rowA) y[ybase + i + q] = rf[q];
                }
            }
        }
    }
    TSPMV_STAMP_STORE
}

// ------------------------------------------------------------------------------------------------
// Column panels (round 4; DevStream::panel_off): y += A_p x for the entries of one run of column panels.  The merged lists of the workgroup entry mode are in column order,
// so a panel of a few MB of x is a run of a list (offsets recorded at plan time); k_units walks the first run with the units, and every further run is one launch of this kernel: the same groups
// of 16 strips, the same records and trip routine, row sums in the same LDS slabs, then y += for the strips that have entries.  Why launches: on scattered matrices with
// a large x every gather misses the XCD's L2 and pulls a 128-byte line across the fabric (profiles/r04_pmc_*_before.json); the workgroups would have to sweep x TOGETHER
// for the lines to be shared, and a kernel boundary is the one chip-wide synchronisation that costs microseconds instead of the polls and timetables of S6.17 — inside one
// launch every gather of the chip falls into one panel.  The price is the read-modify-write of y per pass.  A piece of a split tile-row adds its sums to ITS slot of the
// partial sums (nobody else writes that slot, and the passes are launches in stream order); launch_entry_panels runs k_fixup_split over all split rows behind the last pass, which
// adds the slots up in slot order — so a panelled plan's sums are as reproducible as the plain launch's.  (Rounds 4-5 added the pieces into y atomically: on R-MAT 22 x 8, 4,048 split
// rows, two runs of ONE plan differed in 5 k rows' last bits while the plan's facts said "ordered" — scripts/archive/rounds/r5b_repro_check.py.)
// ------------------------------------------------------------------------------------------------
template <bool NTS>
__global__ __launch_bounds__(256, ECOO2_MIN_WAVES) void k_entries_acc(DevStream S, int rowA, int xcd_chunk, int panel, val_t *__restrict__ partial, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    __shared__ lacc_t s_acc[GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16];
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4;
    unsigned bid = blockIdx.x;
    if (xcd_chunk > 0) {   // the same workgroup -> XCD windows as k_units: a group's rows of y and its x neighbourhood stay with the XCD that touched them
        const unsigned C = (unsigned)xcd_chunk, W = 8u * C, win = bid / W, off = bid % W, k = off & 7u;
        if ((win + 1) * W <= gridDim.x) bid = win * W + k * C + (off >> 3);
    }
    const int4 wr = S.wg_coo[bid];
    const int *po = S.panel_off + (size_t)bid * (size_t)(S.x_panels + 1);
    const int gs = po[min(S.x_panels, panel * S.panel_merge)], ge = po[min(S.x_panels, (panel + 1) * S.panel_merge)];   // this pass: panels [panel * merge, (panel + 1) * merge)
    if (ge <= gs) return;   // workgroup-uniform: nothing of this group in this pass
    const long long task_id = (long long)bid * GROUPS_PER_BLOCK + g;
    int4 t0 = make_int4(0, 0, 0, 0), t1 = make_int4(0, -1, 0, 0);
    if (task_id < S.ntasks) {
        t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
        t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    }
    const bool side = t0.w > t0.z;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    // the strip's rows of y are requested now, ahead of the records and the gathers: the pass is a short chain of dependent hops (ranges -> records -> gathers -> y), and this
    // takes the last one out of it
    val_t yold[STRIP_MAX_ROWS];
#pragma unroll
    for (int k = 0; k < STRIP_MAX_ROWS; k++) {
        const long long yi = ((long long)row0 + k) * 16 + r;
        yold[k] = (side && part < 0 && k < nrows && yi < rowA) ? y[yi] : (val_t)0;
    }
    for (int i = tid; i < GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16; i += 256) s_acc[i] = 0;
    __syncthreads();
    wg_entry_trips<WCOO_HEAVY_CT, 256, NTS>(S.grec, S.gbase, wr.z, S.dest_bits, S.coo_ordered != 0, x, s_acc, tid, wr.x, ge, gs);
    __syncthreads();
    if (!side) return;
    const lacc_t *mine = s_acc + g * (STRIP_MAX_ROWS * 16);
    if (part >= 0) {
        val_t *slot = partial + (long long)part * 16 + r;
        *slot = (val_t)((lacc_t)*slot + mine[r]);
    } else {
#pragma unroll
        for (int k = 0; k < STRIP_MAX_ROWS; k++) {
            const long long yi = ((long long)row0 + k) * 16 + r;
            if (k < nrows && yi < rowA) y[yi] = (val_t)((lacc_t)yold[k] + mine[k * 16 + r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Column slices pinned to XCDs (round 4; DevStream::slice_passes; scripts/micro/xcd_columns.hip is the model this was sized on).  A panel pass makes all eight XCDs sweep the
// SAME slice of x, one kernel per slice, and pays a read-modify-write of y per pass.  The other way round: the dispatcher hands workgroup b to XCD b & 7, so workgroup b takes
// group b >> 3's entries of column slice (pass * 8 + (b & 7)): an XCD only ever gathers from its own slice, which stays in its 4-MB L2 for the whole launch, and x crosses the
// fabric once instead of once per XCD.  The eight partial sums of a row meet in y by atomic adds of the rows a workgroup touched (the unit kernel stored y before this launch);
// the order of those adds is not fixed, so this form is only chosen when the caller has not asked for bit-reproducible sums.  (If a driver dispatched differently, only the
// speed would change.)
// ------------------------------------------------------------------------------------------------
template <bool NTS, int CT>
__global__ __launch_bounds__(256, ECOO2_MIN_WAVES) void k_entries_xcd(DevStream S, int rowA, int pass, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    __shared__ lacc_t s_acc[GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16];
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4;
    const unsigned bid = blockIdx.x >> 3, slice = (unsigned)pass * 8u + (blockIdx.x & 7u), slices = 8u * (unsigned)S.slice_passes;
    const int4 wr = S.wg_coo[bid];
    const int *po = S.panel_off + (size_t)bid * (size_t)(S.x_panels + 1);
    const unsigned P = (unsigned)S.x_panels;
    const int gs = po[slice * P / slices], ge = po[(slice + 1u) * P / slices];   // slice s = panels [s P / slices, (s + 1) P / slices): together all of them, each once
    if (ge <= gs) return;   // workgroup-uniform: nothing of this group in this slice
    const long long task_id = (long long)bid * GROUPS_PER_BLOCK + g;
    int4 t0 = make_int4(0, 0, 0, 0), t1 = make_int4(0, -1, 0, 0);
    if (task_id < S.ntasks) {
        t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
        t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    }
    const bool side = t0.w > t0.z;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    if (TSPMV_DIAG_XCD_ZERO) for (int i = tid; i < GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16; i += 256) s_acc[i] = 0;
    __syncthreads();
    if (TSPMV_DIAG_XCD_TRIP) wg_entry_trips<CT, 256, NTS>(S.grec, S.gbase, wr.z, S.dest_bits, false, x, s_acc, tid, wr.x, ge, gs);
    __syncthreads();
    TSPMV_DIAG_XCD_SKIP_ADDS
    if (!side) return;
    const lacc_t *mine = s_acc + g * (STRIP_MAX_ROWS * 16);
    if (part >= 0) {
        const long long yi = (long long)row0 * 16 + r;
        const val_t v = (val_t)mine[r];
        if (yi < rowA && v != (val_t)0) atomicAdd(&y[yi], v);
    } else {
#pragma unroll
        for (int k = 0; k < STRIP_MAX_ROWS; k++) {
            const long long yi = ((long long)row0 + k) * 16 + r;
            const val_t v = (val_t)mine[k * 16 + r];
            if (k < nrows && yi < rowA && v != (val_t)0) atomicAdd(&y[yi], v);
        }
    }
}

hipError_t launch_entry_slices(const DevStream &S, int rowA, const val_t *x, val_t *y, hipStream_t st)
{
    const dim3 grid((unsigned)S.n_groups * 8u), blk(256);
    // records per trip (256 x CT): the smallest trip that takes an average (group, slice) run whole — a second trip for a few stragglers costs a full round of latencies
    // (uniform random 4 M rows, 1,536 records per run: CT 6 0.2815 ms, CT 8 0.2563; 8 M rows in two passes, 768 per run: CT 4 0.700, CT 6 0.706, CT 8 0.716)
    const int ct = S.slice_ct;
    for (int p = 0; p < S.slice_passes; p++) {
        if (ct == 8) { if (S.nt_stream) hipLaunchKernelGGL((k_entries_xcd<true, 8>), grid, blk, 0, st, S, rowA, p, x, y);
                       else hipLaunchKernelGGL((k_entries_xcd<false, 8>), grid, blk, 0, st, S, rowA, p, x, y); }
        else if (ct == 4) { if (S.nt_stream) hipLaunchKernelGGL((k_entries_xcd<true, 4>), grid, blk, 0, st, S, rowA, p, x, y);
                            else hipLaunchKernelGGL((k_entries_xcd<false, 4>), grid, blk, 0, st, S, rowA, p, x, y); }
        else { if (S.nt_stream) hipLaunchKernelGGL((k_entries_xcd<true, 6>), grid, blk, 0, st, S, rowA, p, x, y);
               else hipLaunchKernelGGL((k_entries_xcd<false, 6>), grid, blk, 0, st, S, rowA, p, x, y); }
    }
    return hipGetLastError();
}

hipError_t launch_entry_panels(const DevPlan &P, const DevStream &S, int xcd_remap, int xcd_chunk, const val_t *x, val_t *y, hipStream_t st)
{
    const int passes = (S.x_panels + S.panel_merge - 1) / S.panel_merge, rowA = P.rowA;
    for (int p = 1; p < passes; p++) {
        const dim3 grid((unsigned)S.n_groups), blk(256);
        const int xc = xcd_remap == 2 ? xcd_chunk : 0;   // (0 = workgroups in dispatch order)
        if (S.nt_stream) hipLaunchKernelGGL((k_entries_acc<true>), grid, blk, 0, st, S, rowA, xc, p, P.partial, x, y);
        else hipLaunchKernelGGL((k_entries_acc<false>), grid, blk, 0, st, S, rowA, xc, p, P.partial, x, y);
    }
    // the split rows once more, now that their pieces' slots hold the entries of every panel (slot order: the plan's)
    if (passes > 1 && P.nfix > 0) hipLaunchKernelGGL(k_fixup_split, dim3((P.nfix + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(256), 0, st, P, y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Dense tiles on the matrix cores (reference dense kernel: src/tilespmv_cuda.h:664-710).
// A dense tile is 256 contiguous values (stored in operand order, below); k-step s of
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 with lane group q (= k index) covers tile column
// 4q + s: A[row][k=q] = tile[row][4q+s] (four 128-B runs per load), B[k=q][*] = x[16cb + 4q + s].
// The accumulator D is carried ACROSS the tiles of a tile-row (C-in = previous D), so a row with
// n dense tiles costs 4n MFMAs and one 16-value update of y at its end; summation order is fixed.
// Round 5: one wavefront walks DENSE_ROWS_PER_WAVE consecutive row records as ONE flat run of tiles (the records' tile
// ranges are contiguous by construction: hip_plan_stream.hip checks it) — the next tile's loads are always in flight,
// across row boundaries too, and the row's 16 values of y are requested when the row STARTS, so that the read-modify-write
// at its end waits for nothing.  With one row per wavefront (rounds 1-4) a wavefront of the band matrix lived for five
// tiles: three dependent round trips (record, column blocks, first tile) before it streamed 10 KB, and the y update
// behind the last tile — 5.7 TB/s where the unit kernel reaches 6.9.
// ------------------------------------------------------------------------------------------------
#ifndef DENSE_MIN_WAVES
#define DENSE_MIN_WAVES 7   // waves per SIMD asked of the register allocator for k_dense_mfma (72 VGPRs; at 8 waves = 64 VGPRs it spills 20 bytes)
#endif
constexpr int DENSE_ROWS_PER_WAVE = 8;   // at most; fewer where that would leave the chip short of wavefronts (launch_dense_mfma)
template <bool NTS>   // NTS: the tile values (read once) are loaded nontemporally — plans above 400 MB per launch, as in k_units
__global__ __launch_bounds__(256, DENSE_MIN_WAVES) void k_dense_mfma(DevDense D, int R, int rowA, int colA, val_t *__restrict__ partial,
                                                    const val_t *__restrict__ x, val_t *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (r0 >= D.nrows) return;  // whole wavefronts only
    const int nr = min(R, D.nrows - r0);
    const int4 dr = reinterpret_cast<const int4 *>(D.rows)[r0 + min(lane, nr - 1)];   // lane l < nr holds record l: {tile-row, tile_begin, tile_end, partial slot or -1}
    const int tb = __builtin_amdgcn_readlane(dr.y, 0), te = __builtin_amdgcn_readlane(dr.z, nr - 1), last = te - 1;
    const int kq = lane >> 4, rr0 = lane & 15;
    const long long xlast = (long long)colA - 1;
#if defined(TILESPMV_F32)
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#else
    v4d acc = {0., 0., 0., 0.};
#endif
    // k-step s of lane group q covers tile column c = 4q + s (any bijection of the 16 columns onto
    // (s, q) works as long as A and B agree): the four B values of a lane are then x[16cb + 4q .. +3],
    // 32 contiguous bytes, and its four A values are val[(4q + s) * 16 + row].
    // column blocks: 64 tiles per lane load, broadcast per tile with v_readlane; a run longer than that reloads (wave-uniform, rare)
    int cbase = tb;
    int cbv = D.cb[min(tb + lane, last)];
    auto load_tile = [&](int t, val_t (&a)[4], val_t (&b)[4]) {
        const int tc = min(t, last);
        if (tc - cbase >= 64) { cbase = tc; cbv = D.cb[min(tc + lane, last)]; }
        const int cb = __builtin_amdgcn_readlane(cbv, __builtin_amdgcn_readfirstlane(tc - cbase));
        const long long xb = (long long)cb * 16 + 4 * kq;
        const val_t *tv = D.val + (long long)tc * 256;   // operand order (dense_slot): fp32 one 16-byte load per lane, fp64 two, each covering whole lines
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = stream_load<NTS>(tv + dense_slot(rr0, 4 * kq + s));
        if (xb + 3 <= xlast) {
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = x[xb + s];   // contiguous: merged into 16-B loads
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) b[s] = x[min(xb + s, xlast)];  // partial last column block (payload is zero there)
        }
    };
    auto mfma4 = [&](const val_t (&a)[4], const val_t (&b)[4]) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
#if defined(TILESPMV_F32)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
#else
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
#endif
        }
    };
    // the rows of D a lane writes: column 0 of D (lanes 0, 16, 32, 48), rows rr(i)
    auto out_row = [&](int i) {
#if defined(TILESPMV_F32)
        return 4 * kq + i;   // f32 C/D map
#else
        return kq + 4 * i;   // f64 C/D map
#endif
    };
    int cur = 0;
    int row = __builtin_amdgcn_readlane(dr.x, 0), row_end = __builtin_amdgcn_readlane(dr.z, 0), part = __builtin_amdgcn_readlane(dr.w, 0);
    val_t yold[4] = {0, 0, 0, 0};
    auto fetch_y = [&]() {   // the row's values of y, requested when the row starts
        if ((lane & 15) == 0 && part < 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const long long yi = (long long)row * 16 + out_row(i); yold[i] = yi < rowA ? y[yi] : (val_t)0; }
        }
    };
    val_t a0[4], b0[4];
    load_tile(tb, a0, b0);
    fetch_y();
    for (int t = tb; t < te; t++) {  // next tile's loads in flight ahead of this tile's MFMAs (unconditional, clamped: exact wait counts)
        val_t a1[4], b1[4];
        load_tile(t + 1, a1, b1);
        mfma4(a0, b0);
        if (t + 1 == row_end) {   // wave-uniform: the row is complete
            if ((lane & 15) == 0) {  // every column of D holds the same 16 results; column 0 writes them
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int rr = out_row(i);
                    if (part >= 0) partial[(long long)part * 16 + rr] = acc[i];
                    else {
                        const long long yi = (long long)row * 16 + rr;
                        if (yi < rowA) y[yi] = yold[i] + acc[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = 0;
            cur++;
            if (cur < nr) {
                row = __builtin_amdgcn_readlane(dr.x, cur); row_end = __builtin_amdgcn_readlane(dr.z, cur); part = __builtin_amdgcn_readlane(dr.w, cur);
                fetch_y();
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) { a0[s] = a1[s]; b0[s] = b1[s]; }
    }
}

hipError_t launch_dense_mfma(const DevDense &D, bool nt_stream, int rowA, int colA, val_t *partial, const val_t *x, val_t *y, hipStream_t st)
{
    if (D.nrows > 0) {
        // row records per wavefront: up to 8, but never so many that the grid falls below ~16 wavefronts per SIMD of the chip (fem6_46: 36 k records — 8 per wavefront left
        // 4.5 wavefronts per SIMD, one round at low occupancy: 4.6 TB/s)
        const int R = std::max(1, std::min(DENSE_ROWS_PER_WAVE, D.nrows / 16384));
        const dim3 grid((unsigned)((D.nrows + 4 * R - 1) / (4 * R)));
        if (nt_stream) hipLaunchKernelGGL(k_dense_mfma<true>, grid, dim3(256), 0, st, D, R, rowA, colA, partial, x, y);
        else hipLaunchKernelGGL(k_dense_mfma<false>, grid, dim3(256), 0, st, D, R, rowA, colA, partial, x, y);
    }
    return hipGetLastError();
}

// ================================================================================================
// Multi-vector form (SpMM): Y[rows][NV] = A * X[cols][NV], both row-major, so the NV
// values of one row of X are one 16/32/64-byte load and the matrix streams (values, descriptors, COO
// entries, tasks) are read once for NV right-hand sides.  Same plan, same units; per lane NV accumulators.
// COO entries of a strip are held in registers (first 16) and scattered per tile-row into a 16 x NV LDS
// slab when that tile-row retires, so the LDS footprint does not grow with the strip's row count.
// ================================================================================================
template <int NV>
struct alignas(NV * sizeof(val_t) >= 16 ? 16 : NV * sizeof(val_t)) MVec { val_t v[NV]; };

// A lane carries NL = min(NV, 2) right-hand sides (16-byte gathers in fp64); Q = NV / NL lane groups of a
// wavefront work on the SAME strip with different slices of the NV vectors, so a wide NV costs no extra
// registers, the stores of one Y row (NV values) come from Q lanes of one store instruction, and a
// wavefront sees the store latency of one strip instead of four (stores retire in order with the loads).
// skip_entries: the plan's merged, column-ordered entry lists are multiplied by k_entries_mv afterwards (entry-dominated plans with the
// workgroup entry mode); this kernel then handles units only and stores zeros for the rows without units.
// CD: dictionary plans (4-B descriptors): the next chunk's words are prefetched (one register), its patterns gathered at the switch.
#ifndef MV_DEFER_F32_NV8
#define MV_DEFER_F32_NV8 0   // fp32 nvec 8: a retired Y row stored behind the next batch's loads won in round 2 (0.494 -> 0.465 ms); with the round-4 kernel it costs 16-24 bytes of scratch and loses (config 4 0.528 vs 0.469 ms, KKT fp32 0.981 vs 0.968: profiles/r04_spmm.txt)
#endif
#ifndef MV_NT_Y
#define MV_NT_Y 1   // streaming stores of Y (as y in k_units)
#endif
#if MV_NT_Y
#define MV_STORE16(w, p) __builtin_nontemporal_store(w, p)
#else
#define MV_STORE16(w, p) (*(p) = (w))
#endif
template <int NVT, bool CD, bool NTS>
__global__ __launch_bounds__(256, MV_MIN_WAVES) void k_units_mv(DevStream S, int rowA, int colA, int xcd_chunk, int skip_entries, int slab_rows, val_t *__restrict__ partial,
                                                  const val_t *__restrict__ X, val_t *__restrict__ Y)
{
    constexpr int NV = NVT < 2 ? NVT : 2;   // vectors per lane
    constexpr int Q = NVT / NV;             // lane groups per strip
    constexpr int STRIPS = GROUPS_PER_BLOCK / Q;
    typedef MVec<NV> vec_t;
#ifndef MV_UBX
#define MV_UBX 1   // value groups per batch of the multi-vector unit loop (diagnostic builds: 2 = twice the units in flight per wavefront, needs MV_MIN_WAVES <= 5)
#endif
    constexpr int UB = UNIT_GROUP * MV_UBX; // MV_UBX 16-byte value loads per batch (2 units fp64, 4 units fp32 each)
    // the next descriptor chunk is prefetched into registers (4 VGPRs) except in the fp64 nvec 4 / 8 variants: there the
    // prefetch pushed the kernel 12 bytes into scratch at 80 VGPRs, and loading the chunk at the switch is 3.4-4.3 % faster
    // (profiles/r03_spmm.txt)
    // (round 6: the fp32 nvec 8 variant with 12-B descriptors spilled the same 12 bytes: same cure; with the derived units' carried x also the fp32 nvec 4 one)
    constexpr bool MV_PREFETCH_DESC = !(sizeof(val_t) == 8 && NVT >= 4) && !(sizeof(val_t) == 4 && NVT >= 4 && !CD);
    __shared__ lacc_t s_c[GROUPS_PER_BLOCK][16][NV];
    __shared__ val_t s_p[GROUPS_PER_BLOCK][16][NV];
    __shared__ uint4 s_d[GROUPS_PER_BLOCK][DCHUNK];
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4;
    unsigned bid = blockIdx.x;
    {
        const unsigned C = (unsigned)xcd_chunk, W = 8u * C, win = bid / W, off = bid % W, k = off & 7u;
        if (C > 0 && (win + 1) * W <= gridDim.x) bid = win * W + k * C + (off >> 3);
    }
    const int q = g % Q;                    // this lane group's slice: vectors q*NV .. q*NV + NV-1
    const long long task_id = (long long)bid * STRIPS + g / Q;
    if (task_id >= S.ntasks) return;
    const int4 t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
    const int4 t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    const int unit_begin = t0.x, unit_end = t0.y, coo_begin = t0.z, coo_end = t0.w;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    const unsigned nounit = (unsigned)t1.z;
    const int ncoo = skip_entries ? 0 : coo_end - coo_begin;
    // Entry slab (plans with more than a handful of entries per tile-row, slab_rows > 0; dynamic LDS [lane group][slab_rows][16][NV]): a strip whose list does not fit the 16 entries that
    // travel with the prologue scatters ALL its entries up front, 4 x 16 per trip with every load of a trip in flight, and a retiring tile-row just reads its 16 sums — instead of walking the
    // list once per tile-row through three dependent loads (KKT stand-in fp32, nvec 8: 1.07 -> see profiles/r03_spmm.txt).
    const bool epre = slab_rows > 0 && ncoo > 16;
    lacc_t (*s_e)[NV] = reinterpret_cast<lacc_t (*)[NV]>(s_dyn) + (size_t)g * slab_rows * 16;
    typedef val_t grp_t __attribute__((ext_vector_type(UNIT_GROUP)));
    const grp_t *__restrict__ ugrp = reinterpret_cast<const grp_t *>(S.uval) + r;  // group layout (hip_plan.hip): one 16-byte load = UNIT_GROUP units
    const vec_t *__restrict__ Xv = reinterpret_cast<const vec_t *>(X) + q;   // row i, slice q: Xv[i * Q]
    vec_t *__restrict__ Yv = reinterpret_cast<vec_t *>(Y) + q;
    const int last = unit_end - 1;
    const int last_grp = unit_begin + (unit_end - 1 - unit_begin) / UNIT_GROUP * UNIT_GROUP;
    auto load_grp = [&](int u, val_t (&out)[UB]) {
#pragma unroll
        for (int gk = 0; gk < UB; gk += UNIT_GROUP) {
            const grp_t pv = stream_load<NTS>(ugrp + (long long)min(u + gk, last_grp) * (16 / UNIT_GROUP));
#pragma unroll
            for (int k = 0; k < UNIT_GROUP; k++) out[gk + k] = pv[k];
        }
    };
    const bool have_units = unit_begin < unit_end;
    const long long xlast = (long long)colA - 1;

    // first 16 COO entries of the strip: loaded and multiplied up front, scattered when their tile-row retires.  The products wait in LDS (s_p, this lane's own slot),
    // not in registers: held across the whole unit loop they were what the allocator spilled at 80 VGPRs (12-36 bytes of scratch, one reload per strip; round 3's
    // "0 B scratch" was true of the dictionary forms in fp64 only)
    unsigned rb0 = 0xFFFFFFFFu;
    if (epre) {
        for (int k = 0; k < nrows; k++) {
#pragma unroll
            for (int j = 0; j < NV; j++) s_e[k * 16 + r][j] = 0;
        }
        wave_lds_fence();
        constexpr int ECT = 4;
        for (int e0 = coo_begin; e0 < coo_end; e0 += 16 * ECT) {
            unsigned rb[ECT]; int cc[ECT]; val_t cv[ECT]; vec_t xx[ECT];
#pragma unroll
            for (int k = 0; k < ECT; k++) {
                const int e = min(e0 + 16 * k + r, coo_end - 1);
                rb[k] = S.crow[e]; cc[k] = S.ccol[e]; cv[k] = S.cval[e];
            }
#pragma unroll
            for (int k = 0; k < ECT; k++) xx[k] = Xv[(long long)cc[k] * Q];
#pragma unroll
            for (int k = 0; k < ECT; k++)
                if (e0 + 16 * k + r < coo_end) {
#pragma unroll
                    for (int j = 0; j < NV; j++) atomicAdd(&s_e[(rb[k] >> 4) * 16 + (rb[k] & 15u)][j], (lacc_t)(cv[k] * xx[k].v[j]));
                }
        }
        wave_lds_fence();
    } else if (r < ncoo) {
        rb0 = S.crow[coo_begin + r];
        const int cc = S.ccol[coo_begin + r];
        const val_t cv = S.cval[coo_begin + r];
        const vec_t xx = Xv[(long long)cc * Q];
#pragma unroll
        for (int j = 0; j < NV; j++) s_p[g][r][j] = cv * xx.v[j];
    }
    uint4 dcur = make_uint4(0u, 0u, 0u, 0u), dnext = dcur;
    const unsigned *__restrict__ udw = reinterpret_cast<const unsigned *>(S.udesc_cb);
    val_t v[UB];
    if (have_units) {
        if constexpr (CD) {
            dcur.x = udw[min(unit_begin + r, last)];
            if (MV_PREFETCH_DESC) dnext.x = udw[min(unit_begin + DCHUNK + r, last)];
        } else {
            dcur = load_udesc(S.udesc_cb, min(unit_begin + r, last));
            if (MV_PREFETCH_DESC) dnext = load_udesc(S.udesc_cb, min(unit_begin + DCHUNK + r, last));
        }
        load_grp(unit_begin, v);
    }
    if (ncoo > 0 && !epre) {
#pragma unroll
        for (int j = 0; j < NV; j++) s_c[g][r][j] = 0;
        wave_lds_fence();
    }
    val_t acc[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) acc[j] = 0;

    // COO contributions of tile-row kr (of the strip) -> acc.  The strip's list is in tile-row order (the plan appends it
    // tile-row by tile-row), so rows retired in ascending order (`in_order`: the unit loop) continue where the previous row
    // stopped and stop at the first chunk that already holds a later row; rows without units, flushed at the end, scan it all.
    int scan_from = coo_begin + 16;
    auto coo_add = [&](int kr, bool in_order) {
        if (ncoo == 0) return;
        if (epre) {
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = (val_t)((lacc_t)acc[j] + s_e[kr * 16 + r][j]);
            return;
        }
        if ((rb0 >> 4) == (unsigned)kr) {
#pragma unroll
            for (int j = 0; j < NV; j++) atomicAdd(&s_c[g][rb0 & 15u][j], (lacc_t)s_p[g][r][j]);
        }
        for (int e0 = in_order ? scan_from : coo_begin + 16; e0 < coo_end; e0 += 16) {
            unsigned rb = 0xFFFFFFFFu;
            if (e0 + r < coo_end) {
                rb = S.crow[e0 + r];
                if ((rb >> 4) == (unsigned)kr) {
                    const val_t cv = S.cval[e0 + r];
                    const vec_t xx = Xv[(long long)S.ccol[e0 + r] * Q];
#pragma unroll
                    for (int j = 0; j < NV; j++) atomicAdd(&s_c[g][rb & 15u][j], (lacc_t)(cv * xx.v[j]));
                }
            }
            if (in_order) {
                scan_from = e0;   // the next row starts looking here
                const unsigned long long later = __ballot(rb != 0xFFFFFFFFu && (rb >> 4) > (unsigned)kr);
                if ((later >> (tid & 48)) & 0xFFFFull) break;   // this chunk already holds a later tile-row: nothing of kr beyond it
            }
        }
        wave_lds_fence();
        // (the zeros that reset the slab come from a move the optimiser cannot hoist: kept live across the unit loop as a 16-byte constant they were spilled and RELOADED FROM
        //  SCRATCH here in the fp32 build — 20 bytes of scratch for four zeros)
        unsigned zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int j = 0; j < NV; j++) {
            acc[j] = (val_t)((lacc_t)acc[j] + s_c[g][r][j]);
            *reinterpret_cast<uint2 *>(&s_c[g][r][j]) = make_uint2(zero, zero);
        }
        wave_lds_fence();
    };
    // Stores count in the same in-order vmcnt queue as loads: a store issued at the end of a batch makes the next batch's wait
    // for its gathers also wait for the store's acknowledge.  Where registers allow (no spill at 6 waves/SIMD), a retired row
    // waits in registers and is stored right AFTER the next batch's loads have been issued: nvec 2 fp64 0.256 -> 0.249 ms, nvec 8
    // fp32 0.494 -> 0.465 ms; with the spills it brings elsewhere (fp64 nvec 4 / 8: 6 VGPRs) it loses 5-7 %, so it is per variant.
    constexpr bool MV_DEFER_STORE = NVT == 2 || (MV_DEFER_F32_NV8 && sizeof(val_t) == 4 && NVT == 8);
    vec_t pend; int pend_kr = -1;
    auto store_vec = [&](int kr, const vec_t &o) {
        const long long yi = ((long long)row0 + kr) * 16 + r;
        if (yi < rowA) {
            if constexpr (sizeof(vec_t) == 16) {
                v4u_t w; __builtin_memcpy(&w, &o, 16);
                MV_STORE16(w, reinterpret_cast<v4u_t *>(&Yv[yi * Q]));
            } else if constexpr (sizeof(vec_t) == 8) {
                v2u_t w; __builtin_memcpy(&w, &o, 8);
                __builtin_nontemporal_store(w, reinterpret_cast<v2u_t *>(&Yv[yi * Q]));
            } else {
                Yv[yi * Q] = o;
            }
        }
    };
    auto flush_pending = [&]() { if (pend_kr >= 0) { store_vec(pend_kr, pend); pend_kr = -1; } };
    auto store_row = [&](int kr) {
        const long long yi = ((long long)row0 + kr) * 16 + r;
        if (MV_DEFER_STORE) {
            flush_pending();
#pragma unroll
            for (int j = 0; j < NV; j++) { pend.v[j] = acc[j]; acc[j] = 0; }
            pend_kr = kr;
            return;
        }
        if (yi < rowA) {
            vec_t o;
#pragma unroll
            for (int j = 0; j < NV; j++) o.v[j] = acc[j];
            if constexpr (sizeof(vec_t) == 16) {  // streaming stores, as in k_units
                v4u_t w; __builtin_memcpy(&w, &o, 16);
                MV_STORE16(w, reinterpret_cast<v4u_t *>(&Yv[yi * Q]));
            } else if constexpr (sizeof(vec_t) == 8) {
                v2u_t w; __builtin_memcpy(&w, &o, 8);
                __builtin_nontemporal_store(w, reinterpret_cast<v2u_t *>(&Yv[yi * Q]));
            } else {
                Yv[yi * Q] = o;
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) acc[j] = 0;
    };

    if (have_units) {
        const uint2 *sd = reinterpret_cast<const uint2 *>(&s_d[g][0]) + (r >> 3);
        if constexpr (CD) s_d[g][r] = udesc_expand(S, dcur.x, udict_of(S, dcur.x));
        else s_d[g][r] = dcur;
        wave_lds_fence();
        int chunk_end = unit_begin + DCHUNK;
        val_t xprev[NV];   // the x the previous unit used (derived units, unit_x_use)
#pragma unroll
        for (int j = 0; j < NV; j++) xprev[j] = 0;
        for (int u = unit_begin; u < unit_end; u += UB) {
            if (u == chunk_end) {
                wave_lds_fence();
                if constexpr (CD) {
                    const unsigned w = MV_PREFETCH_DESC ? dnext.x : udw[min(chunk_end + r, last)];
                    s_d[g][r] = udesc_expand(S, w, udict_of(S, w));
                } else s_d[g][r] = MV_PREFETCH_DESC ? dnext : load_udesc(S.udesc_cb, min(chunk_end + r, last));
                wave_lds_fence();
                chunk_end += DCHUNK;
                if constexpr (CD) { if (MV_PREFETCH_DESC) dnext.x = udw[min(chunk_end + r, last)]; }
                else if (MV_PREFETCH_DESC) dnext = load_udesc(S.udesc_cb, min(chunk_end + r, last));
            }
            const int j0 = u - (chunk_end - DCHUNK);
            uint2 d[UB];
            vec_t xv[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) d[k] = sd[2 * (j0 + k)];
#pragma unroll
            for (int k = 0; k < UB; k++) {
                const unsigned fl = d[k].x >> 24;
                const unsigned nib = (fl & UNIT_ROWUNIT) ? (unsigned)r : (d[k].y >> (28 - 4 * (r & 7))) & 15u;
                xv[k] = Xv[min(unit_x_base(d[k].x) + nib, xlast) * Q];
            }
            val_t vn[UB];
            load_grp(u + UB, vn);
            if (MV_DEFER_STORE) flush_pending();   // the row retired by the previous batch: behind this batch's loads
#pragma unroll
            for (int k = 0; k < UB; k++) {
                if (u + k >= unit_end) break;
                const unsigned fl = d[k].x >> 24;
                if (fl & UNIT_ROWUNIT) {
                    const bool target = r == (int)(d[k].y & 15u);
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        const val_t sum = strip_allreduce(v[k] * xv[k].v[j]);
                        if (target) acc[j] += sum;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NV; j++) {   // (derived units: the previous unit's x one lane up, unit_x_use)
                        const val_t xu = unit_x_use(xv[k].v[j], xprev[j], d[k].x, r);
                        xprev[j] = xu;
                        acc[j] += v[k] * xu;
                    }
                }
                if (fl & UNIT_EOR) {
                    const int kr = (int)((fl >> UNIT_ROW_SHIFT) & 7u);
                    coo_add(kr, true);
                    store_row(kr);
                }
            }
#pragma unroll
            for (int k = 0; k < UB; k++) v[k] = vn[k];
        }
    }
    if (MV_DEFER_STORE) flush_pending();
    if (part >= 0) {  // piece of a split tile-row: its partial sums go to the slot, k_fixup_split_mv adds the slots up
        coo_add(0, false);
        vec_t o;
#pragma unroll
        for (int j = 0; j < NV; j++) o.v[j] = acc[j];
        reinterpret_cast<vec_t *>(partial)[((long long)part * 16 + r) * Q + q] = o;
    } else {
        unsigned m = nounit;  // tile-rows without any unit: COO contributions only (or zero)
        while (m) { const int kr = __ffs((int)m) - 1; m &= m - 1; coo_add(kr, false); store_row(kr); }
        if (MV_DEFER_STORE) flush_pending();
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-vector product on POOLED plans (round 5; hip_plan.h "pooled units"): Y[rows][NVT] = A X[cols][NVT].  Same division of labour as k_units_mv — a lane carries NV = min(NVT, 2)
// right-hand sides, Q = NVT / NV lane groups of a wavefront work on the SAME strip with different slices of the vectors — and the pooled kernel's data flow: the strip's slab
// [tile-row in strip][row][NVT] (fp64 sums, 16 KB per workgroup whatever NVT) is zeroed, the strip's list entries are scattered into it up front, every unit adds its 16 products per
// right-hand side with ds_add (unconditionally: a unit past the task's end adds 0), and the rows of Y are stored from the slab.  The matrix streams are read once for all NVT vectors;
// before this kernel a pooled plan ran SpMM one right-hand side at a time.
// ------------------------------------------------------------------------------------------------
template <int NVT, bool NTS, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? 5 : MV_MIN_WAVES) void k_pool_mv(DevStream S, int rowA, int colA, int xcd_chunk, val_t *__restrict__ partial, const val_t *__restrict__ X, val_t *__restrict__ Y)
{
    constexpr int NV = NVT < 2 ? NVT : 2;   // vectors per lane
    constexpr int Q = NVT / NV;             // lane groups per strip
    constexpr int STRIPS = GROUPS_PER_BLOCK / Q;
    constexpr int SR = POOL_STRIP_ROWS;
    typedef MVec<NV> vec_t;
    constexpr int UB = UNIT_GROUP;
    // slab: [tile-row in strip][vector][row], the 16 rows of one vector contiguous (the lanes of a strip hit neighbouring banks) and the vectors 20 doubles apart, so that the lane
    // groups of a wavefront — same rows, different vectors — do not all fall into the same banks (16 apart = 128 bytes = all of them into one half of the banks)
    constexpr int VS = 20;
    __shared__ lacc_t s_acc[STRIPS][SR][NVT][VS];
    __shared__ uint4 s_d[GROUPS_PER_BLOCK][DCHUNK];
    typedef typename std::conditional<WIDE, uint4, uint2>::type side_t;   // per parked unit: its row nibbles (16-column pooled plans) or its 16 column-offset bytes (wide pooled plans, S.ucol: +2 KB of LDS, one workgroup fewer per CU)
    __shared__ side_t s_r[GROUPS_PER_BLOCK][DCHUNK];
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4;
    unsigned bid = blockIdx.x;
    {
        const unsigned C = (unsigned)xcd_chunk, W = 8u * C, win = bid / W, off = bid % W, k = off & 7u;
        if (C > 0 && (win + 1) * W <= gridDim.x) bid = win * W + k * C + (off >> 3);
    }
    const int q = g % Q, sg = g / Q;        // this lane group's slice of the vectors, its strip inside the workgroup
    const long long task_id = (long long)bid * STRIPS + sg;
    if (task_id >= S.ntasks) return;        // (no workgroup barrier below: lane groups leave on their own)
    const int4 t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
    const int4 t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    const int unit_begin = t0.x, unit_end = t0.y, coo_begin = t0.z, coo_end = t0.w;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    typedef val_t grp_t __attribute__((ext_vector_type(UNIT_GROUP)));
    const grp_t *__restrict__ ugrp = reinterpret_cast<const grp_t *>(S.uval) + r;
    const uint2 *__restrict__ urw = reinterpret_cast<const uint2 *>(S.urow);
    const vec_t *__restrict__ Xv = reinterpret_cast<const vec_t *>(X) + q;   // row i, slice q: Xv[i * Q]
    vec_t *__restrict__ Yv = reinterpret_cast<vec_t *>(Y) + q;
    const int last = unit_end - 1;
    const int last_grp = unit_begin + (unit_end - 1 - unit_begin) / UNIT_GROUP * UNIT_GROUP;
    const bool have_units = unit_begin < unit_end;
    const long long xlast = (long long)colA - 1;
    lacc_t (*slab)[NVT][VS] = s_acc[sg];
    // the Q lane groups of a strip zero their own slices of the slab: [k][r][q NV .. q NV + NV)
    for (int k = 0; k < nrows; k++) {
#pragma unroll
        for (int j = 0; j < NV; j++) slab[k][q * NV + j][r] = 0;
    }
    uint4 dcur = make_uint4(0u, 0u, 0u, 0u), dnext = dcur;
    side_t rcur{}, rnext{};   // row nibbles — or, wide pooled plans, the 16 column-offset bytes
    constexpr bool wide = WIDE;   // (wide pooled plans: the descriptor's nibble words are the row nibbles, S.ucol the column offsets)
    auto side_of = [&](int i) -> side_t { if constexpr (WIDE) return S.ucol[i]; else return urw[i]; };
    val_t v[UB];
    auto load_grp = [&](int u, val_t (&out)[UB]) {
        const grp_t pv = stream_load<NTS>(ugrp + (long long)min(u, last_grp) * (16 / UNIT_GROUP));
#pragma unroll
        for (int k = 0; k < UNIT_GROUP; k++) out[k] = pv[k];
    };
    // pooled dictionary plans (S.pdict): 8-byte descriptors (word 0, pattern id); the pattern — column and row nibbles — is gathered one chunk after its id was loaded
    const bool pd = S.pdict != nullptr;
    uint2 wnn2 = make_uint2(0u, 0u);
    if (have_units) {   // descriptor chunks 0 and 1, first value group: in flight across the entry phase
        if (pd) {
            const uint2 a = pool_desc(S, min(unit_begin + r, last)), b = pool_desc(S, min(unit_begin + DCHUNK + r, last));
            dcur.x = a.x; dcur.y = a.y; dnext.x = b.x; dnext.y = b.y;
        } else {
            dcur = load_udesc_raw(S.udesc, min(unit_begin + r, last)); dnext = load_udesc_raw(S.udesc, min(unit_begin + DCHUNK + r, last));
            rcur = side_of(min(unit_begin + r, last)); rnext = side_of(min(unit_begin + DCHUNK + r, last));
        }
        load_grp(unit_begin, v);
    }
    wave_lds_fence();
    // ---- the strip's list entries: every load of a trip in flight, then the gathers, then the adds (each lane group for its own slice of the vectors)
    {
        constexpr int ECT = 4;
        for (int e0 = coo_begin; e0 < coo_end; e0 += 16 * ECT) {
            unsigned rb[ECT]; int cc[ECT]; val_t cv[ECT]; vec_t xx[ECT];
#pragma unroll
            for (int k = 0; k < ECT; k++) {
                const int e = min(e0 + 16 * k + r, coo_end - 1);
                rb[k] = S.crow[e]; cc[k] = S.ccol[e]; cv[k] = S.cval[e];
            }
#pragma unroll
            for (int k = 0; k < ECT; k++) xx[k] = Xv[(long long)cc[k] * Q];
#pragma unroll
            for (int k = 0; k < ECT; k++)
                if (e0 + 16 * k + r < coo_end) {
#pragma unroll
                    for (int j = 0; j < NV; j++) atomicAdd(&slab[rb[k] >> 4][q * NV + j][rb[k] & 15u], (lacc_t)(cv[k] * xx[k].v[j]));
                }
        }
    }
    // ---- units
    if (have_units) {
        const uint2 *sd = reinterpret_cast<const uint2 *>(&s_d[g][0]) + (r >> 3);
        const unsigned *sr = reinterpret_cast<const unsigned *>(&s_r[g][0]) + (r >> 3);            // this lane's half of a unit's row nibbles
        const unsigned char *sc = reinterpret_cast<const unsigned char *>(&s_r[g][0]) + r;          // wide: this lane's byte of a unit's column offsets
        if (pd) {
            const uint4 p0 = S.pdict[dcur.y], p1 = S.pdict[dnext.y];
            wnn2 = pool_desc(S, min(unit_begin + 2 * DCHUNK + r, last));
            dcur = make_uint4(dcur.x, p0.x, p0.y, 0u); dnext = make_uint4(dnext.x, p1.x, p1.y, 0u);
            if constexpr (!WIDE) { rcur = make_uint2(p0.z, p0.w); rnext = make_uint2(p1.z, p1.w); }
        }
        s_d[g][r] = udesc_park_form(dcur); s_r[g][r] = rcur;
        wave_lds_fence();
        int chunk_end = unit_begin + DCHUNK;
        for (int u = unit_begin; u < unit_end; u += UB) {
            if (u == chunk_end) {
                wave_lds_fence();
                s_d[g][r] = udesc_park_form(dnext); s_r[g][r] = rnext;
                wave_lds_fence();
                chunk_end += DCHUNK;
                if (pd) {
                    const uint4 p = S.pdict[wnn2.y];
                    dnext = make_uint4(wnn2.x, p.x, p.y, 0u);
                    if constexpr (!WIDE) rnext = make_uint2(p.z, p.w);
                    wnn2 = pool_desc(S, min(chunk_end + DCHUNK + r, last));
                } else { dnext = load_udesc_raw(S.udesc, min(chunk_end + r, last)); rnext = side_of(min(chunk_end + r, last)); }
            }
            const int j0 = u - (chunk_end - DCHUNK);
            uint2 d[UB]; unsigned rw[UB]; vec_t xv[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) { d[k] = sd[2 * (j0 + k)]; rw[k] = wide ? d[k].y : sr[2 * (j0 + k)]; }
#pragma unroll
            for (int k = 0; k < UB; k++) {
                const unsigned coff = wide ? (unsigned)sc[16 * (j0 + k)] : (d[k].y >> (28 - 4 * (r & 7))) & 15u;
                xv[k] = Xv[min((long long)(d[k].x & POOL_BASE_MASK) + (long long)coff, xlast) * Q];
            }
            val_t vn[UB];
            load_grp(u + UB, vn);
#pragma unroll
            for (int k = 0; k < UB; k++) {
                const val_t vk = (u + k < unit_end) ? v[k] : (val_t)0;
                const unsigned kr = (d[k].x >> POOL_KR_SHIFT) & 7u, rn = (rw[k] >> (28 - 4 * (r & 7))) & 15u;
#pragma unroll
                for (int j = 0; j < NV; j++) atomicAdd(&slab[kr][q * NV + j][rn], (lacc_t)(vk * xv[k].v[j]));
            }
#pragma unroll
            for (int k = 0; k < UB; k++) v[k] = vn[k];
        }
    }
    wave_lds_fence();
    // ---- results: this lane group's slice of the strip's rows
    if (part >= 0) {
        vec_t o;
#pragma unroll
        for (int j = 0; j < NV; j++) o.v[j] = (val_t)slab[0][q * NV + j][r];
        reinterpret_cast<vec_t *>(partial)[((long long)part * 16 + r) * Q + q] = o;
    } else {
        for (int k = 0; k < nrows; k++) {
            const long long yi = ((long long)row0 + k) * 16 + r;
            if (yi < rowA) {
                vec_t o;
#pragma unroll
                for (int j = 0; j < NV; j++) o.v[j] = (val_t)slab[k][q * NV + j][r];
                if constexpr (sizeof(vec_t) == 16) { v4u_t w; __builtin_memcpy(&w, &o, 16); MV_STORE16(w, reinterpret_cast<v4u_t *>(&Yv[yi * Q])); }
                else if constexpr (sizeof(vec_t) == 8) { v2u_t w; __builtin_memcpy(&w, &o, 8); __builtin_nontemporal_store(w, reinterpret_cast<v2u_t *>(&Yv[yi * Q])); }
                else Yv[yi * Q] = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-vector entry pass (round 3): Y[rows][NVT] += A_entries * X for plans whose COO entries run per workgroup (entry mode 2, 16 strips):
// the same merged, column-ordered, packed lists k_units<.., 2> walks, two right-hand sides per pass (one 16-byte gather per entry and lane
// in fp64), NVT / 2 passes over the workgroup's list (the list of a workgroup is ~80 KB: the later passes find it in L2).  Row sums of
// a pass are accumulated in LDS (2,048 rows x 2 values) exactly like the single-vector kernel does, then added to Y, which k_units_mv
// (skip_entries) has written before.  Pieces of split tile-rows add their sums to their own slot of the partial sums (k_fixup_split_mv runs behind this pass).  Entry-dominated plans used to run one
// right-hand side at a time below nvec 8 (webbase-like: 40 / 83 us for nvec 2 / 4).
// ------------------------------------------------------------------------------------------------
template <int NVT>
__global__ __launch_bounds__(256) void k_entries_mv(DevStream S, int rowA, val_t *__restrict__ partial, const val_t *__restrict__ X, val_t *__restrict__ Y)
{
    constexpr int NP = NVT / 2, CT = 4;
    typedef MVec<2> vec_t;
    __shared__ lacc_t s_acc[GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16][2];
    const int tid = threadIdx.x, r = tid & 15, g = tid >> 4, wave = tid >> 6;
    const unsigned bid = blockIdx.x;
    const int4 wr = S.wg_coo[bid];
    if (wr.y <= wr.x) return;   // workgroup-uniform
    const long long task_id = (long long)bid * GROUPS_PER_BLOCK + g;
    int4 t0 = make_int4(0, 0, 0, 0), t1 = make_int4(0, -1, 0, 0);
    if (task_id < S.ntasks) {
        t0 = reinterpret_cast<const int4 *>(S.task)[task_id * 2];
        t1 = reinterpret_cast<const int4 *>(S.task)[task_id * 2 + 1];
    }
    const bool side = t0.w > t0.z;
    const int row0 = t1.x, part = t1.y, nrows = t1.w;
    const int db = S.dest_bits, gb = wr.x, ge = wr.y;
    const unsigned dmask = (1u << db) - 1u;
    const int clast = wr.z + ((ge - 1 - gb) >> 6);
    const bool ordered = S.coo_ordered != 0;
    const vec_t *__restrict__ X2 = reinterpret_cast<const vec_t *>(X);
    vec_t *__restrict__ Y2 = reinterpret_cast<vec_t *>(Y);
    for (int p = 0; p < NP; p++) {
        for (int i = tid; i < GROUPS_PER_BLOCK * STRIP_MAX_ROWS * 16; i += 256) { s_acc[i][0] = 0; s_acc[i][1] = 0; }
        __syncthreads();
        for (int e0 = gb; e0 < ge; e0 += 256 * CT) {
            ERec rr[CT]; unsigned cb[CT]; vec_t xx[CT];
#pragma unroll
            for (int q = 0; q < CT; q++) {
                rr[q] = S.grec[min(e0 + 256 * q + tid, ge - 1)];
                cb[q] = S.gbase[__builtin_amdgcn_readfirstlane(min(wr.z + ((e0 - gb) >> 6) + 4 * q + wave, clast))];
            }
#pragma unroll
            for (int q = 0; q < CT; q++) xx[q] = X2[(size_t)(cb[q] + (rr[q].w >> db)) * NP + p];
            auto adds = [&]() {
#pragma unroll
                for (int q = 0; q < CT; q++)
                    if (e0 + 256 * q + tid < ge) {
                        const val_t v = erec_val(rr[q]);
                        atomicAdd(&s_acc[rr[q].w & dmask][0], (lacc_t)(v * xx[q].v[0]));
                        atomicAdd(&s_acc[rr[q].w & dmask][1], (lacc_t)(v * xx[q].v[1]));
                    }
            };
            if (ordered) {   // wavefronts add in turn: plan-fixed order of the additions, as in k_units<.., 2>
                for (int w = 0; w < 4; w++) { if (wave == w) adds(); __syncthreads(); }
            } else adds();
        }
        __syncthreads();
        if (side) {
            if (part >= 0) {   // piece of a split tile-row: the entry sums join the piece's own slot; k_fixup_split_mv adds the slots up afterwards, in slot order (no atomics: reproducible)
                val_t *slot = partial + (((long long)part * 16 + r) * NP + p) * 2;
                slot[0] = (val_t)((lacc_t)slot[0] + s_acc[g * (STRIP_MAX_ROWS * 16) + r][0]);
                slot[1] = (val_t)((lacc_t)slot[1] + s_acc[g * (STRIP_MAX_ROWS * 16) + r][1]);
            } else {
                for (int k = 0; k < nrows; k++) {
                    const long long yi = ((long long)row0 + k) * 16 + r;
                    if (yi < rowA) {
                        vec_t o = Y2[yi * NP + p];
                        o.v[0] = (val_t)((lacc_t)o.v[0] + s_acc[g * (STRIP_MAX_ROWS * 16) + k * 16 + r][0]);
                        o.v[1] = (val_t)((lacc_t)o.v[1] + s_acc[g * (STRIP_MAX_ROWS * 16) + k * 16 + r][1]);
                        Y2[yi * NP + p] = o;
                    }
                }
            }
        }
        __syncthreads();
    }
}

template <int NV>
__global__ __launch_bounds__(256) void k_fixup_split_mv(DevPlan P, val_t *__restrict__ Y)
{
    const int f_id = blockIdx.x * GROUPS_PER_BLOCK + (threadIdx.x >> 4), r = threadIdx.x & 15;
    if (f_id >= P.nfix) return;
    const FixRow f = P.fix[f_id];
    val_t sum[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) sum[j] = 0;
    for (int k = 0; k < f.count; k++)
#pragma unroll
        for (int j = 0; j < NV; j++) sum[j] += P.partial[((long long)(f.first + k) * 16 + r) * NV + j];
    const long long yi = (long long)f.row * 16 + r;
    if (yi < P.rowA)
#pragma unroll
        for (int j = 0; j < NV; j++) Y[yi * NV + j] = sum[j];
}

// Dense tiles, multi-vector: the B operand finally is a matrix — B[k][n] = X[16 cb + col(k)][n] for the
// n < NV right-hand sides (zero beyond), D[row][n] accumulates across the tiles of the tile-row as before.
template <int NV>
__global__ __launch_bounds__(256) void k_dense_mfma_mv(DevDense D, int rowA, int colA, val_t *__restrict__ partial,
                                                       const val_t *__restrict__ X, val_t *__restrict__ Y)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= D.nrows) return;
    const int4 dr = reinterpret_cast<const int4 *>(D.rows)[w];
    const int row = dr.x, t0 = dr.y, t1 = dr.z, part = dr.w;
    const int last = t1 - 1, kq = lane >> 4, n = lane & 15;
    const long long xlast = (long long)colA - 1;
#if defined(TILESPMV_F32)
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#else
    v4d acc = {0., 0., 0., 0.};
#endif
    const int cbv = D.cb[min(t0 + lane, last)];
    for (int t = t0; t < t1; t++) {
        const int cb = __builtin_amdgcn_readlane(cbv, __builtin_amdgcn_readfirstlane(t - t0));
        const long long xb = (long long)cb * 16 + 4 * kq;
        const val_t *tv = D.val + (long long)t * 256;
        val_t a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = tv[dense_slot(n, 4 * kq + s)];
#pragma unroll
        for (int s = 0; s < 4; s++) b[s] = n < NV ? X[min(xb + s, xlast) * NV + n] : (val_t)0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
#if defined(TILESPMV_F32)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
#else
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
#endif
        }
    }
    if (n < NV) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
#if defined(TILESPMV_F32)
            const int rr = 4 * kq + i;
#else
            const int rr = kq + 4 * i;
#endif
            if (part >= 0) partial[((long long)part * 16 + rr) * NV + n] = acc[i];
            else {
                const long long yi = (long long)row * 16 + rr;
                if (yi < rowA) Y[yi * NV + n] += acc[i];
            }
        }
    }
}

template <int NV>
static hipError_t launch_mv(const DevPlan &P, const DevStream &S, const DevDense &DN, int xcd_chunk, bool entries_pass, int slab_rows, const val_t *X, val_t *Y, hipStream_t st)
{
    if (S.ntasks > 0)
    {
        constexpr int strips = GROUPS_PER_BLOCK / (NV < 2 ? 1 : NV / 2);  // per workgroup (k_units_mv: Q lane groups per strip)
        const int slab = entries_pass ? 0 : slab_rows;   // (the entry pass multiplies the lists itself)
        const size_t slab_bytes = (size_t)GROUPS_PER_BLOCK * slab * 16 * (NV < 2 ? NV : 2) * sizeof(lacc_t);
#define TSPMV_MV(CD, NTS) hipLaunchKernelGGL((k_units_mv<NV, CD, NTS>), dim3((unsigned)((S.ntasks + strips - 1) / strips)), dim3(256), slab_bytes, st, \
                                             S, P.rowA, P.colA, xcd_chunk, entries_pass ? 1 : 0, slab, P.partial, X, Y)
        if (S.pooled) {   // pooled plans: their own multi-vector kernel (entries in-kernel: no entry pass, no dynamic slab)
            const dim3 grid((unsigned)((S.ntasks + strips - 1) / strips));
            if (S.ucol) { if (S.nt_stream) hipLaunchKernelGGL((k_pool_mv<NV, true, true>), grid, dim3(256), 0, st, S, P.rowA, P.colA, xcd_chunk, P.partial, X, Y);
                          else hipLaunchKernelGGL((k_pool_mv<NV, false, true>), grid, dim3(256), 0, st, S, P.rowA, P.colA, xcd_chunk, P.partial, X, Y); }
            else if (S.nt_stream) hipLaunchKernelGGL((k_pool_mv<NV, true, false>), grid, dim3(256), 0, st, S, P.rowA, P.colA, xcd_chunk, P.partial, X, Y);
            else hipLaunchKernelGGL((k_pool_mv<NV, false, false>), grid, dim3(256), 0, st, S, P.rowA, P.colA, xcd_chunk, P.partial, X, Y);
        } else if (S.cb_bits > 0) { if (S.nt_stream) TSPMV_MV(true, true); else TSPMV_MV(true, false); }
        else { if (S.nt_stream) TSPMV_MV(false, true); else TSPMV_MV(false, false); }
#undef TSPMV_MV
    }
    if (DN.nrows > 0)
        hipLaunchKernelGGL((k_dense_mfma_mv<NV>), dim3((DN.nrows + 3) / 4), dim3(256), 0, st, DN, P.rowA, P.colA, P.partial, X, Y);
    if (entries_pass && !S.pooled && S.ntasks > 0)   // Y += entries (after the units and the dense pass have written Y; pieces of split rows: into their slots)
        hipLaunchKernelGGL((k_entries_mv<NV>), dim3((unsigned)((S.ntasks + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK)), dim3(256), 0, st, S, P.rowA, P.partial, X, Y);
    if (P.nfix > 0)   // the split rows last: slot order
        hipLaunchKernelGGL((k_fixup_split_mv<NV>), dim3((P.nfix + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(256), 0, st, P, Y);
    return hipGetLastError();
}

// nvec in {2, 4, 8}.  The plan must be a unit-stream plan without whole-tile passes and without the CSR
// fallback (the defaults); the caller checks that (hip_plan.hip).
hipError_t launch_tiles_stream_mv(const DevPlan &P, const DevStream &S, const DevDense &DN, int nvec, int xcd_chunk, bool entries_pass, int slab_rows, const val_t *X, val_t *Y,
                                  hipStream_t st)
{
    switch (nvec) {
    case 2: return launch_mv<2>(P, S, DN, xcd_chunk, entries_pass, slab_rows, X, Y, st);
    case 4: return launch_mv<4>(P, S, DN, xcd_chunk, entries_pass, slab_rows, X, Y, st);
    case 8: return launch_mv<8>(P, S, DN, xcd_chunk, entries_pass, slab_rows, X, Y, st);
    default: return hipErrorInvalidValue;
    }
}

// ---- multi-vector product for plans without a native multi-vector kernel (generation-1 plans, whole CSR tiles, CSR
// fallback) and for plans whose work is mostly COO entries: one right-hand side at a time through the plan's own SpMV.  The
// row-major X is transposed into nvec contiguous vectors first and the results are transposed back into Y afterwards (one
// pass each: a thread moves one row, so both sides of both kernels are coalesced).
__global__ __launch_bounds__(256) void k_rows_to_columns(const val_t *__restrict__ X, int nvec, long long n, long long ld, val_t *__restrict__ XT)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int j = 0; j < nvec; j++) XT[j * ld + i] = X[i * nvec + j];
}
__global__ __launch_bounds__(256) void k_columns_to_rows(const val_t *__restrict__ YT, int nvec, long long row0, long long rows, long long ld, val_t *__restrict__ Y)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    for (int j = 0; j < nvec; j++) Y[(row0 + i) * nvec + j] = YT[j * ld + i];
}
// Vectors between the caller's numbering and a reordered plan's (host_reorder.cpp: perm[new] = old): gather out[i] = in[perm[i]] (x into plan order), scatter out[perm[i]] = in[i]
// (y back).  16 bytes of perm and of the contiguous side per 4 lanes: the scattered side is what it costs (one 128-byte line per element at worst).
__global__ __launch_bounds__(256) void k_permute_vector(const val_t *__restrict__ in, val_t *__restrict__ out, const int *__restrict__ perm, long long n, int scatter)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long p = perm[i];
    if (scatter) out[p] = in[i]; else out[i] = in[p];
}
hipError_t launch_permute_vector(const val_t *in, val_t *out, const int *perm, long long n, int scatter, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_permute_vector, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, perm, n, scatter);
    return hipGetLastError();
}

hipError_t launch_rows_to_columns(const val_t *X, int nvec, long long n, long long ld, val_t *XT, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_rows_to_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, nvec, n, ld, XT);
    return hipGetLastError();
}
hipError_t launch_columns_to_rows(const val_t *YT, int nvec, long long row0, long long rows, long long ld, val_t *Y, hipStream_t st)
{
    if (rows > 0) hipLaunchKernelGGL(k_columns_to_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, YT, nvec, row0, rows, ld, Y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Plan creation on the device, first piece (round 5; SURVEY S8 f1 "device-side ..."): the ENCODE stage's value pass.  The builder emits the 16 values of every unit in tile order
// (src); what the unit kernel reads is, per task, groups of UNIT_GROUP units interleaved per row (hip_plan.h "unit stream").  On the host that permutation is a second full pass
// over the plan's largest array (0.67 GB for config 4) into a staging copy; here the emitted values are uploaded as they are and one workgroup per task writes them to their final
// place.  map[i] = {first unit of task i in src, first unit in dst, units, -}.  Padding slots of a task's last group are never written: the arena block they lie in was zeroed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pair_values(const val_t *__restrict__ src, val_t *__restrict__ dst, const int4 *__restrict__ map, int ntasks)
{
    constexpr int G = UNIT_GROUP;
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const int4 m = map[t];
        const long long so = (long long)m.x * 16, dofs = (long long)m.y * 16;
        for (int i = threadIdx.x; i < m.z * 16; i += 256) {
            const int j = i >> 4, r = i & 15;
            dst[dofs + (long long)(j / G * G) * 16 + G * r + (j % G)] = src[so + i];
        }
    }
}

hipError_t launch_pair_values(const val_t *src, val_t *dst, const int4 *map, int ntasks)
{
    if (ntasks > 0) hipLaunchKernelGGL(k_pair_values, dim3((unsigned)std::min(ntasks, 1 << 20)), dim3(256), 0, nullptr, src, dst, map, ntasks);
    return hipGetLastError();
}

hipError_t launch_tiles_stream(const DevPlan &P, const DevStream &S, const DevDense &DN, bool dense_mfma, int entry_mode, int wg_strips, int lds_pad_bytes, int xcd_remap, int xcd_chunk,
                               const val_t *x, val_t *y, hipStream_t st)
{
    if (S.ntasks > 0) {
        const int xc = xcd_remap == 2 ? xcd_chunk : 0;   // workgroup -> XCD windows: a kernel argument (0 = dispatch order)
        // k_units<UB, entry mode, strips per workgroup, dictionary descriptors, nontemporal streams, pooled, wide>
#define TSPMV_K(W, B, CD, NTS, PL, WD) hipLaunchKernelGGL((k_units<TILESPMV_UB, W, B, CD, NTS, PL, WD>), dim3((unsigned)((S.ntasks + B - 1) / B)), dim3(16 * B), (size_t)lds_pad_bytes, st, S, P.rowA, P.colA, xc, P.partial, x, y)
#define TSPMV_NTS(W, B, CD, PL, WD) do { if (S.nt_stream) TSPMV_K(W, B, CD, true, PL, WD); else TSPMV_K(W, B, CD, false, PL, WD); } while (0)
        // the descriptor form of the plan: classic 12-B / dictionary; pooled 20-B / pooled dictionary / wide
#define TSPMV_FORM_(W, B, L) do { \
            if (S.pooled) { if (S.ucol) L(W, B, false, true, true); else if (S.pdict) L(W, B, true, true, false); else L(W, B, false, true, false); } \
            else if (S.cb_bits > 0) L(W, B, true, false, false); else L(W, B, false, false, false); } while (0)
#define TSPMV_PLAIN(W, B, CD, PL, WD) TSPMV_K(W, B, CD, false, PL, WD)
#define TSPMV_FORM(W, B) TSPMV_FORM_(W, B, TSPMV_NTS)          /* large plans: nontemporal streams where the plan says so */
#define TSPMV_FORM_SMALL(W, B) TSPMV_FORM_(W, B, TSPMV_PLAIN)  /* small grids, entry mode 1: default cache policy */
        // Grids that would give fewer than half the CUs a 256-thread workgroup run 128-thread workgroups of 8 strips instead: twice the workgroups, the same strips
        // (per-strip and per-wavefront entry modes only: the workgroup mode merges the lists of its 16 strips at plan creation)
        static const int small_grid_workgroups = [] { const char *e = getenv("TILESPMV_SMALL_GRID_WORKGROUPS"); return e && *e ? atoi(e) : SMALL_GRID_WORKGROUPS; }();   // (0 switches the form off)
        const bool small_grid = entry_mode != 2 && !S.nt_stream && (S.ntasks + 15) / 16 < small_grid_workgroups;
        if (small_grid) { if (entry_mode == 1) TSPMV_FORM_SMALL(1, 8); else TSPMV_FORM_SMALL(0, 8); }
        else if (entry_mode == 1) TSPMV_FORM_SMALL(1, 16);                                    // (entry mode 1 = small grids: never nontemporal)
        else if (entry_mode == 2 && wg_strips == 32 && !S.pooled) {                          // 512-thread workgroups: classic / dictionary descriptors only
            if (S.cb_bits > 0) TSPMV_NTS(2, 32, true, false, false); else TSPMV_NTS(2, 32, false, false, false);
        }
        else if (entry_mode == 2) TSPMV_FORM(2, 16);
        else TSPMV_FORM(0, 16);
#undef TSPMV_FORM_SMALL
#undef TSPMV_FORM
#undef TSPMV_PLAIN
#undef TSPMV_FORM_
#undef TSPMV_NTS
#undef TSPMV_K
    }
    // whole-tile passes (y += ...): CSR tiles kept as tiles, dense tiles on the matrix cores; then the split-row fix-up
    hipError_t e = launch_tiles_direct(P, dense_mfma, /*accumulate=*/true, /*fixup=*/false, x, y, st);
    if (e != hipSuccess) return e;
    e = launch_dense_mfma(DN, S.nt_stream != 0, P.rowA, P.colA, P.partial, x, y, st);
    if (e != hipSuccess) return e;
    if (P.nfix_late > 0) {  // split rows with pieces outside k_units (whole-tile / matrix-core passes)
        DevPlan Q = P;
        Q.fix = P.fix_late; Q.nfix = P.nfix_late;
        hipLaunchKernelGGL(k_fixup_split, dim3((Q.nfix + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK), dim3(256), 0, st, Q, y);
    }
    if (S.slice_passes > 0 && S.x_panels > 1 && S.ntasks > 0) return launch_entry_slices(S, P.rowA, x, y, st);   // the lists, by column slices pinned to XCDs
    if (S.panel_merge > 0 && S.x_panels > S.panel_merge && S.ntasks > 0) return launch_entry_panels(P, S, xcd_remap, xcd_chunk, x, y, st);   // y += the entries of the other column panels, last: every row has been written
    return hipGetLastError();
}

}  // namespace tilespmv
