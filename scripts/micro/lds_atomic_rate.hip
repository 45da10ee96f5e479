// lds_atomic_rate.hip — how fast are LDS floating-point atomic adds on gfx950, by element type and address pattern?
// (The fp32 build of the entry phase spends 0.108 ms of a 0.166 ms SpMV in its ds_add_f32; the fp64 build's ds_add_f64 are free.)
// One workgroup of 256 threads per CU x 4, each lane performs N adds into a 2048-element LDS array:
//   pattern 0: lane-private elements (no two lanes share an address or a bank)
//   pattern 1: pseudo-random elements (the SpMV's scatter)
//   pattern 2: all lanes of a 16-lane group add to one element
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/lds_atomic_rate.hip -o /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class T, int PATTERN, bool AS_INT>
__global__ __launch_bounds__(256) void k(T *out, int n)
{
    __shared__ T s[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s[i] = 0;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    const T one = (T)1;
    for (int i = 0; i < n; i++) {
        h = h * 1664525u + 1013904223u;
        const unsigned idx = PATTERN == 0 ? (threadIdx.x + 256u * (i & 7)) : PATTERN == 1 ? (h >> 21) : ((threadIdx.x >> 4) * 16 + (i & 15) * 128) & 2047u;
        if constexpr (AS_INT) atomicAdd(reinterpret_cast<unsigned *>(&s[idx]), 1u);
        else atomicAdd(&s[idx], one);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5];
}

template <class T, int PATTERN, bool AS_INT> static int run(const char *name)
{
    T *out; CK(hipMalloc(&out, 4096 * sizeof(T)));
    const int wgs = 1024, n = 4096;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<T, PATTERN, AS_INT>), dim3(wgs), dim3(256), 0, 0, out, n); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k<T, PATTERN, AS_INT>), dim3(wgs), dim3(256), 0, 0, out, n);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)wgs * 256 * n;
    printf("%-34s %8.3f ms  %7.1f G adds/s  (%.2f ns per wavefront instruction per CU)\n", name, ms, adds / ms * 1e-6, ms * 1e6 / (adds / 64 / 256));
    CK(hipFree(out));
    return 0;
}

int main()
{
    run<float, 0, false>("f32 add, lane-private"); run<double, 0, false>("f64 add, lane-private"); run<float, 0, true>("u32 add, lane-private");
    run<float, 1, false>("f32 add, random"); run<double, 1, false>("f64 add, random"); run<float, 1, true>("u32 add, random");
    run<float, 2, false>("f32 add, 16 lanes per element"); run<double, 2, false>("f64 add, 16 lanes per element"); run<float, 2, true>("u32 add, 16 lanes per element");
    return 0;
}
