// Microbenchmark: which read pattern / concurrency reaches the HBM rate on MI355X?
// Build: hipcc -O3 --offload-arch=gfx950 stream_patterns.hip -o stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: every wave reads a contiguous chunk, 8 B per lane per load, UB loads in flight.
template <int UB, int MINW>
__global__ __launch_bounds__(256, MINW) void k_wave_contig(const double *__restrict__ a, double *__restrict__ out, long long n_per_wave)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const double *p = a + wave * n_per_wave;
    double acc = 0;
    for (long long i = 0; i < n_per_wave; i += 64 * UB) {
        double v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = p[i + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k];
    }
    if (acc == 123.456) out[0] = acc;
}
// B: every 16-lane group reads its own contiguous strip, 128 B per load, UB loads in flight (the tile kernel's pattern).
template <int UB, int MINW>
__global__ __launch_bounds__(256, MINW) void k_group_strips(const double *__restrict__ a, double *__restrict__ out, long long n_per_group)
{
    const int r = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const double *p = a + grp * n_per_group;
    double acc = 0;
    for (long long i = 0; i < n_per_group; i += 16 * UB) {
        double v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = p[i + 16 * k + r];
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k];
    }
    if (acc == 123.456) out[0] = acc;
}
// C: like A with 16 B per lane.
template <int UB, int MINW>
__global__ __launch_bounds__(256, MINW) void k_wave_contig16(const double2 *__restrict__ a, double *__restrict__ out, long long n_per_wave)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const double2 *p = a + wave * n_per_wave;
    double acc = 0;
    for (long long i = 0; i < n_per_wave; i += 64 * UB) {
        double2 v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = p[i + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k].x + v[k].y;
    }
    if (acc == 123.456) out[0] = acc;
}

// D: group strips, but LANEB bytes per lane (more instructions for the same bytes) -> instruction-rate ceiling
template <int UB, class T>
__global__ __launch_bounds__(256) void k_group_small(const T *__restrict__ a, double *__restrict__ out, long long n_per_group)
{
    const int r = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const T *p = a + grp * n_per_group;
    T acc = 0;
    for (long long i = 0; i < n_per_group; i += 16 * UB) {
        T v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = p[i + 16 * k + r];
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k];
    }
    if (acc == (T)123) out[0] = (double)acc;
}
// E: like B plus a second, small "descriptor" stream: 8 B per lane, 2 distinct addresses per 16 lanes, 16 B per unit
template <int UB>
__global__ __launch_bounds__(256) void k_group_desc(const double *__restrict__ a, const uint2 *__restrict__ dsc, double *__restrict__ out, long long n_per_group)
{
    const int r = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const double *p = a + grp * n_per_group;
    const uint2 *q = dsc + grp * (n_per_group / 16) * 2 + (r >> 3);
    double acc = 0;
    for (long long i = 0; i < n_per_group; i += 16 * UB) {
        double v[UB]; uint2 d[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) { d[k] = q[(i / 16 + k) * 2]; v[k] = p[i + 16 * k + r]; }
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k] * (double)(d[k].x + d[k].y);
    }
    if (acc == 123.456) out[0] = acc;
}

// F: descriptor interleaved with the payload: 144-byte unit records [16 doubles][16 B descriptor]
template <int UB, int RECB>
__global__ __launch_bounds__(256) void k_group_rec(const char *__restrict__ a, double *__restrict__ out, long long units_per_group)
{
    const int r = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const char *p = a + grp * units_per_group * RECB;
    double acc = 0;
    for (long long i = 0; i < units_per_group; i += UB) {
        double v[UB]; uint2 d[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const char *rec = p + (i + k) * RECB;
            v[k] = *reinterpret_cast<const double *>(rec + 8 * r);
            d[k] = *reinterpret_cast<const uint2 *>(rec + 128 + 8 * (r >> 3));
        }
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k] * (double)(d[k].x + d[k].y);
    }
    if (acc == 123.456) out[0] = acc;
}

// G: E (payload + descriptor stream, 20 units per lane group) plus what a strip writes: 4 rows x 128 B per group,
// either as four 8-B-per-lane stores or as two 16-B-per-lane stores, or none
template <int UB, int STORE>
__global__ __launch_bounds__(256) void k_group_desc_store(const double *__restrict__ a, const uint2 *__restrict__ dsc, double *__restrict__ y, long long n_per_group)
{
    const int r = threadIdx.x & 15;
    const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const double *p = a + grp * n_per_group;
    const uint2 *q = dsc + grp * (n_per_group / 16) * 2 + (r >> 3);
    double acc = 0;
    for (long long i = 0; i < n_per_group; i += 16 * UB) {
        double v[UB]; uint2 d[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) { d[k] = q[(i / 16 + k) * 2]; v[k] = p[i + 16 * k + r]; }
#pragma unroll
        for (int k = 0; k < UB; k++) acc += v[k] * (double)(d[k].x + d[k].y);
    }
    double *yo = y + grp * 64;
    if (STORE == 1) { for (int k = 0; k < 4; k++) yo[16 * k + r] = acc + k; }
    else if (STORE == 2) { double2 t0 = {acc, acc + 1}, t1 = {acc + 2, acc + 3}; reinterpret_cast<double2 *>(yo)[r] = t0; reinterpret_cast<double2 *>(yo)[16 + r] = t1; }
    else if (acc == 123.456) yo[0] = acc;
}

template <class F> double timeit(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}

int main()
{
    const long long N = 1ll << 27;  // 128 Mi doubles = 1 GiB
    double *a, *out;
    CK(hipMalloc(&a, N * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 0, N * 8));
    const double gb = N * 8 / 1e9;
#define RUNA(UB, MINW, PERWAVE) { long long pw = PERWAVE; long long waves = N / pw; dim3 g((unsigned)(waves / 4)); \
    double ms = timeit([&] { hipLaunchKernelGGL((k_wave_contig<UB, MINW>), g, dim3(256), 0, 0, a, out, pw); }); \
    printf("A wave-contig   8B/lane UB=%d minw=%d bytes/wave=%6lld : %.4f ms  %.0f GB/s\n", UB, MINW, pw * 8, ms, gb / ms * 1e3); }
#define RUNB(UB, MINW, PERGRP) { long long pg = PERGRP; long long grps = N / pg; dim3 g((unsigned)(grps / 16)); \
    double ms = timeit([&] { hipLaunchKernelGGL((k_group_strips<UB, MINW>), g, dim3(256), 0, 0, a, out, pg); }); \
    printf("B group-strips  8B/lane UB=%d minw=%d bytes/group=%6lld : %.4f ms  %.0f GB/s\n", UB, MINW, pg * 8, ms, gb / ms * 1e3); }
#define RUNC(UB, MINW, PERWAVE) { long long pw = PERWAVE; long long waves = N / 2 / pw; dim3 g((unsigned)(waves / 4)); \
    double ms = timeit([&] { hipLaunchKernelGGL((k_wave_contig16<UB, MINW>), g, dim3(256), 0, 0, (const double2 *)a, out, pw); }); \
    printf("C wave-contig  16B/lane UB=%d minw=%d bytes/wave=%6lld : %.4f ms  %.0f GB/s\n", UB, MINW, pw * 16, ms, gb / ms * 1e3); }
    RUNA(1, 1, 2048) RUNA(2, 1, 2048) RUNA(4, 1, 2048) RUNA(8, 1, 2048) RUNA(4, 1, 8192) RUNA(8, 1, 8192) RUNA(8, 1, 32768) RUNA(16, 1, 32768)
    RUNA(4, 5, 2048) RUNA(4, 5, 8192)
    RUNB(1, 1, 512) RUNB(2, 1, 512) RUNB(4, 1, 512) RUNB(8, 1, 512) RUNB(4, 1, 2048) RUNB(8, 1, 2048) RUNB(16, 1, 2048) RUNB(4, 5, 512) RUNB(4, 5, 2048) RUNB(8, 4, 2048)
#define RUND(UB, T, PERGRPBYTES) { long long pg = PERGRPBYTES / sizeof(T); long long grps = N * 8 / PERGRPBYTES; dim3 g((unsigned)(grps / 16)); \
    double ms = timeit([&] { hipLaunchKernelGGL((k_group_small<UB, T>), g, dim3(256), 0, 0, (const T *)a, out, pg); }); \
    printf("D group-strips %zuB/lane UB=%d bytes/group=%6d : %.4f ms  %.0f GB/s  (%.1f G wave-loads/s)\n", sizeof(T), UB, PERGRPBYTES, ms, gb / ms * 1e3, (double)N * 8 / sizeof(T) / 64 / ms * 1e-6); }
    RUND(4, float, 4096) RUND(8, float, 4096) RUND(4, short, 4096) RUND(8, short, 4096) RUND(16, short, 4096)
    { uint2 *dsc; hipMalloc(&dsc, N / 16 * 16); hipMemset(dsc, 0, N / 16 * 16);
      long long pg = 512; long long grps = N / pg; dim3 g((unsigned)(grps / 16));
      double ms = timeit([&] { hipLaunchKernelGGL((k_group_desc<4>), g, dim3(256), 0, 0, a, dsc, out, pg); });
      printf("E group-strips 8B/lane + 16B/unit descriptor stream UB=4 bytes/group=4096: %.4f ms  %.0f GB/s (payload+desc)\n", ms, (gb * 1.125) / ms * 1e3);
      pg = 320; grps = N / pg; g = dim3((unsigned)(grps / 16));
      ms = timeit([&] { hipLaunchKernelGGL((k_group_desc<4>), g, dim3(256), 0, 0, a, dsc, out, pg); });
      printf("E same, bytes/group=2560 (20 units, like a Laplacian strip): %.4f ms  %.0f GB/s\n", ms, (gb * 1.125) / ms * 1e3); }
#define RUNF(UB, RECB, UPG) { long long upg = UPG; long long grps = (N * 8) / (upg * RECB); grps -= grps % 16; dim3 g((unsigned)(grps / 16)); \
    double ms = timeit([&] { hipLaunchKernelGGL((k_group_rec<UB, RECB>), g, dim3(256), 0, 0, (const char *)a, out, upg); }); \
    printf("F unit records %dB (payload+desc interleaved) UB=%d units/group=%d : %.4f ms  %.0f GB/s\n", RECB, UB, UPG, ms, (double)grps * upg * RECB / 1e9 / ms * 1e3); }
    RUNF(4, 144, 20) RUNF(4, 144, 32) RUNF(2, 144, 20) RUNF(8, 144, 32) RUNF(4, 160, 20) RUNF(4, 256, 20) RUNF(5, 144, 20)
    { uint2 *dsc; hipMalloc(&dsc, N / 16 * 16); hipMemset(dsc, 0, N / 16 * 16); double *yy; hipMalloc(&yy, N / 320 * 64 * 8 + 4096);
      long long pg = 320, grps = N / pg; grps -= grps % 16; dim3 g((unsigned)(grps / 16));
      double m0 = timeit([&] { hipLaunchKernelGGL((k_group_desc_store<4, 0>), g, dim3(256), 0, 0, a, dsc, yy, pg); });
      double m1 = timeit([&] { hipLaunchKernelGGL((k_group_desc_store<4, 1>), g, dim3(256), 0, 0, a, dsc, yy, pg); });
      double m2 = timeit([&] { hipLaunchKernelGGL((k_group_desc_store<4, 2>), g, dim3(256), 0, 0, a, dsc, yy, pg); });
      printf("G 20-unit strips + desc: no store %.4f ms | 4x8B-lane stores %.4f ms | 2x16B-lane stores %.4f ms  (reads %.0f MB, writes %.0f MB)\n", m0, m1, m2, gb * 1125, (double)grps * 512 / 1e6); }
    RUNC(1, 1, 1024) RUNC(2, 1, 1024) RUNC(4, 1, 1024) RUNC(4, 1, 4096) RUNC(8, 1, 4096)
    return 0;
}
