// nt_stream_bw.hip — what a streaming kernel reaches on this card with default-policy and with nontemporal loads / stores
// (the ceiling to hold k_units against once its value stream is read nontemporally).
//   read-only      : every lane loads 16 B per step, sums; 1 GiB, 2 GiB working sets (> the 256 MB Infinity Cache)
//   read + write   : the SpMV-like mix — 7 parts read, 1 part written (a second buffer), same policies
// Each figure: best of 5 runs of 3 back-to-back launches (hipEvents).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/nt_stream_bw.hip -o scripts/micro/nt_stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ v4u ld(const v4u *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(v4u *p, v4u v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// a workgroup owns a contiguous slab of `per_wg` 16-byte elements; 8 loads in flight per lane
template <bool NTL>
__global__ __launch_bounds__(256) void k_read(const v4u *__restrict__ a, unsigned *__restrict__ out, long long per_wg)
{
    const v4u *p = a + (long long)blockIdx.x * per_wg + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
    for (long long i = 0; i < per_wg; i += 256 * 8) {
        v4u v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = ld<NTL>(p + i + 256 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

// 7 slabs read, the element-wise sum of every 7 consecutive chunks written: reads : writes = 7 : 1
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_read_write(const v4u *__restrict__ a, v4u *__restrict__ w, long long per_wg)
{
    const v4u *p = a + (long long)blockIdx.x * per_wg * 7 + threadIdx.x;
    v4u *q = w + (long long)blockIdx.x * per_wg + threadIdx.x;
    for (long long i = 0; i < per_wg; i += 256) {
        v4u v[7];
#pragma unroll
        for (int k = 0; k < 7; k++) v[k] = ld<NTL>(p + (i * 7) + 256 * k);
        v4u s = v[0];
#pragma unroll
        for (int k = 1; k < 7; k++) s += v[k];
        st<NTS>(q + i, s);
    }
}

template <class F> static float best_ms(F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a, 0); for (int k = 0; k < 3; k++) launch(); hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = std::min(best, ms / 3);
    }
    return best;
}

int main()
{
    unsigned *out; CK(hipMalloc(&out, 64));
    for (long long gib : {1ll, 2ll}) {
        const long long bytes = gib << 30, n16 = bytes / 16;
        v4u *a, *w; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&w, bytes / 7 + (1 << 20)));
        CK(hipMemset(a, 1, bytes));
        for (int wgs : {2048, 8192, 32768}) {
            const long long per = n16 / wgs / (256 * 8) * (256 * 8);
            const double gb = (double)per * wgs * 16 / 1e9;
            const float m0 = best_ms([&] { hipLaunchKernelGGL(k_read<false>, dim3(wgs), dim3(256), 0, 0, a, out, per); });
            const float m1 = best_ms([&] { hipLaunchKernelGGL(k_read<true>, dim3(wgs), dim3(256), 0, 0, a, out, per); });
            printf("read-only  %lld GiB, %5d workgroups: default %.4f ms %5.0f GB/s | nontemporal %.4f ms %5.0f GB/s\n", gib, wgs, m0, gb / m0 * 1e3, m1, gb / m1 * 1e3);
        }
        for (int wgs : {8192, 32768}) {
            const long long per = n16 / 7 / wgs / 256 * 256;
            const double gb = (double)per * wgs * 16 * 8 / 1e9;   // 7 read + 1 written
            const float m00 = best_ms([&] { hipLaunchKernelGGL((k_read_write<false, false>), dim3(wgs), dim3(256), 0, 0, a, w, per); });
            const float m01 = best_ms([&] { hipLaunchKernelGGL((k_read_write<false, true>), dim3(wgs), dim3(256), 0, 0, a, w, per); });
            const float m10 = best_ms([&] { hipLaunchKernelGGL((k_read_write<true, false>), dim3(wgs), dim3(256), 0, 0, a, w, per); });
            const float m11 = best_ms([&] { hipLaunchKernelGGL((k_read_write<true, true>), dim3(wgs), dim3(256), 0, 0, a, w, per); });
            printf("read 7 : write 1, %lld GiB read, %5d workgroups (GB/s of read + written bytes): default/default %5.0f | default loads, nt stores %5.0f | nt loads, default stores %5.0f | nt/nt %5.0f\n",
                   gib, wgs, gb / m00 * 1e3, gb / m01 * 1e3, gb / m10 * 1e3, gb / m11 * 1e3);
        }
        CK(hipFree(a)); CK(hipFree(w));
    }
    return 0;
}
