// How fast can the CUs read a buffer that fits the 256 MiB Infinity Cache (vs one that does not)?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UB>
__global__ __launch_bounds__(256) void k_read(const double *__restrict__ a, double *__restrict__ out, long long n_per_group)
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
int main()
{
    double *a, *out; hipMalloc(&a, 1ll << 31); hipMalloc(&out, 64); hipMemset(a, 0, 1ll << 31);
    for (long long mib : {16ll, 32ll, 64ll, 128ll, 192ll, 256ll, 512ll, 1024ll, 2048ll}) {
        const long long N = mib * 1024 * 1024 / 8, pg = 512, grps = N / pg;
        dim3 g((unsigned)(grps / 16));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_read<4>), g, dim3(256), 0, 0, a, out, pg);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_read<4>), g, dim3(256), 0, 0, a, out, pg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("re-read %5lld MiB: %.4f ms  %.0f GB/s\n", mib, ms, (double)N * 8 / 1e9 / ms * 1e3);
    }
    return 0;
}
