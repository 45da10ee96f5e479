// How many bytes cross the fabric per scattered 8-byte gather, by the KIND of memory the table lives in?  (round 4: the irregular class pays a 128-byte line per gather from
// ordinary hipMalloc memory whatever the load's cache bits; does uncached / fine-grained memory change the granule?)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_granule.hip -o scripts/micro/gather_granule
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- scripts/micro/gather_granule        (separate passes for TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int KIND>   // the kernel name carries the memory kind so that counter rows can be told apart
__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ t, unsigned long long mask, int per_thread, double *__restrict__ out)
{
    unsigned long long s = (unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    double acc = 0;
    for (int i = 0; i < per_thread; i += 4) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { s = s * 6364136223846793005ull + 1442695040888963407ull; v[q] = t[(s >> 20) & mask]; }
#pragma unroll
        for (int q = 0; q < 4; q++) acc += v[q];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int KIND>
static void run(const char *name, double *t, size_t n, double *out)
{
    const int blocks = 256 * 24, per_thread = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_gather<KIND>, dim3(blocks), dim3(256), 0, 0, t, (unsigned long long)(n - 1), per_thread, out);
    hipEventRecord(a);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_gather<KIND>, dim3(blocks), dim3(256), 0, 0, t, (unsigned long long)(n - 1), per_thread, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double g = (double)blocks * 256 * per_thread;
    printf("%-34s table %4zu MB: %.3f ms per launch, %.1f G gathers/s (%.0f M gathers per launch)\n", name, n * 8 >> 20, ms, g / ms * 1e-6, g * 1e-6);
}

int main()
{
    double *out; hipMalloc(&out, 64);
    for (size_t mb : {64, 512}) {
        const size_t n = mb << 17;   // doubles
        std::vector<double> h(n, 1.0);
        double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
        if (hipMalloc(&t0, n * 8) == hipSuccess) { hipMemcpy(t0, h.data(), n * 8, hipMemcpyHostToDevice); run<0>("hipMalloc", t0, n, out); hipFree(t0); }
        if (hipExtMallocWithFlags((void **)&t1, n * 8, hipDeviceMallocUncached) == hipSuccess) { hipMemcpy(t1, h.data(), n * 8, hipMemcpyHostToDevice); run<1>("hipDeviceMallocUncached", t1, n, out); hipFree(t1); }
        else printf("hipDeviceMallocUncached: not available\n");
        if (hipExtMallocWithFlags((void **)&t2, n * 8, hipDeviceMallocFinegrained) == hipSuccess) { hipMemcpy(t2, h.data(), n * 8, hipMemcpyHostToDevice); run<2>("hipDeviceMallocFinegrained", t2, n, out); hipFree(t2); }
        else printf("hipDeviceMallocFinegrained: not available\n");
    }
    return 0;
}
