// hip_prims.hip — the one translation unit that instantiates rocPRIM (hip_prims.h).  Value-type independent: the fp64 and fp32 libraries compile the same code.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <algorithm>
#include <rocprim/iterator/transform_iterator.hpp>

#include "hip_prims.h"

namespace tilespmv {
namespace prims {

// ---- exclusive scan of ints, hand-written (rocPRIM's lookback scan is 1.1 MB of code objects for what is a few dozen microseconds of the builders' time): blocks of SCAN_TILE
// elements are summed, the block sums scanned by the same routine one level up (17 M elements: 8,192 sums, then 4), and every block scans its tile again from its offset.
// in == out is allowed: a thread reads its eight elements before it writes them, and nobody else touches them.
constexpr int SCAN_THREADS = 256, SCAN_PER_THREAD = 8, SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;

template <bool APPLY>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(const int *__restrict__ in, int *__restrict__ out, int *__restrict__ sums, size_t n)
{
    __shared__ int s_wave[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)tid * SCAN_PER_THREAD;
    int v[SCAN_PER_THREAD], mine = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) { v[k] = base + k < n ? in[base + k] : 0; mine += v[k]; }
    int incl = mine;                       // inclusive scan of the threads' sums: inside the wavefront by shuffles, across the four wavefronts through LDS
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if constexpr (!APPLY) {
        if (tid == SCAN_THREADS - 1) sums[blockIdx.x] = before + incl;
    } else {
        int run = (sums ? sums[blockIdx.x] : 0) + before + incl - mine;
#pragma unroll
        for (int k = 0; k < SCAN_PER_THREAD; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    }
}

static size_t scan_levels_ints(size_t n)   // ints of scratch: the block sums of every level
{
    size_t total = 0;
    while (n > (size_t)SCAN_TILE) { n = (n + SCAN_TILE - 1) / SCAN_TILE; total += (n + 63) / 64 * 64; }
    return total;
}

static hipError_t scan_level(int *scratch, const int *in, int *out, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 1) {
        hipLaunchKernelGGL((k_scan_tiles<true>), dim3(1), dim3(SCAN_THREADS), 0, st, in, out, (int *)nullptr, n);
        return hipGetLastError();
    }
    int *sums = scratch;
    hipLaunchKernelGGL((k_scan_tiles<false>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, (int *)nullptr, sums, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = scan_level(scratch + (nb + 63) / 64 * 64, sums, sums, nb, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_scan_tiles<true>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, sums, n);
    return hipGetLastError();
}

hipError_t scan_int(void *tmp, size_t &tmp_b, const int *in, int *out, size_t n, hipStream_t st)
{
    if (tmp == nullptr) { tmp_b = std::max<size_t>(16, scan_levels_ints(n) * sizeof(int)); return hipSuccess; }
    if (n > ((size_t)1 << 42)) return hipErrorInvalidValue;
    return scan_level((int *)tmp, in, out, n, st);
}

template <class K, class V>
static hipError_t sort_pairs(void *tmp, size_t &tmp_b, K *&k_cur, K *&k_alt, V *&v_cur, V *&v_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st)
{
    rocprim::double_buffer<K> kb(k_cur, k_alt);
    rocprim::double_buffer<V> vb(v_cur, v_alt);
    const hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_b, kb, vb, n, b0, b1, st);
    if (tmp != nullptr && e == hipSuccess) { k_cur = kb.current(); k_alt = kb.alternate(); v_cur = vb.current(); v_alt = vb.alternate(); }
    return e;
}

hipError_t sort_pairs_u64_int(void *tmp, size_t &tmp_b, u64 *&k_cur, u64 *&k_alt, int *&v_cur, int *&v_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st)
{
    return sort_pairs<u64, int>(tmp, tmp_b, k_cur, k_alt, v_cur, v_alt, n, b0, b1, st);
}

hipError_t sort_pairs_u32_int(void *tmp, size_t &tmp_b, unsigned *&k_cur, unsigned *&k_alt, int *&v_cur, int *&v_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st)
{
    return sort_pairs<unsigned, int>(tmp, tmp_b, k_cur, k_alt, v_cur, v_alt, n, b0, b1, st);
}

hipError_t sort_keys_u64(void *tmp, size_t &tmp_b, u64 *&k_cur, u64 *&k_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st)
{
    rocprim::double_buffer<u64> kb(k_cur, k_alt);
    const hipError_t e = rocprim::radix_sort_keys(tmp, tmp_b, kb, n, b0, b1, st);
    if (tmp != nullptr && e == hipSuccess) { k_cur = kb.current(); k_alt = kb.alternate(); }
    return e;
}

struct ShiftKey {
    unsigned shift;
    __host__ __device__ u64 operator()(u64 k) const { return k >> shift; }
};

hipError_t rle_u64(void *tmp, size_t &tmp_b, const u64 *in, unsigned shift, unsigned n, u64 *uniq, int *counts, int *nruns, hipStream_t st)
{
    auto it = rocprim::make_transform_iterator(in, ShiftKey{shift});
    return rocprim::run_length_encode(tmp, tmp_b, it, n, uniq, counts, nruns, st);
}

}  // namespace prims
}  // namespace tilespmv
