// Do physical chunks of the card's memory differ in speed?  (round 4, DESIGN.md S6.19: identical plans run in one of two states 13 % apart depending on where their blocks land.)
// N physical chunks (hipMemCreate) of C MB, each mapped at its own virtual range; every pass streams through ALL of them once (N x C far above the 256-MB Infinity Cache), one
// kernel launch per chunk, timed per chunk with events; the per-chunk mean over the passes is printed sorted, with the chunk's index in allocation order.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/chunk_bw.hip -o scripts/micro/chunk_bw;   scripts/micro/chunk_bw [chunk MB = 256] [chunks = 48] [passes = 12]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_stream(const u4 *__restrict__ p, size_t n16, unsigned *__restrict__ sink)
{
    u4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u4 v = __builtin_nontemporal_load(p + i); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 256;
    const int N = argc > 2 ? atoi(argv[2]) : 48, passes = argc > 3 ? atoi(argv[3]) : 12;
    const size_t bytes = mb << 20;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<void *> va(N);
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    for (int i = 0; i < N; i++) {
        CK(hipMemCreate(&h[i], bytes, &prop, 0));
        CK(hipMemAddressReserve(&va[i], bytes, 0, nullptr, 0));
        CK(hipMemMap(va[i], bytes, 0, h[i], 0));
        CK(hipMemSetAccess(va[i], bytes, &acc, 1));
        CK(hipMemset(va[i], i + 1, bytes));
    }
    unsigned *sink; CK(hipMalloc(&sink, 64));
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> e0(N), e1(N);
    for (int i = 0; i < N; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    std::vector<double> sum(N, 0), mn(N, 1e9);
    for (int p = -1; p < passes; p++) {
        for (int i = 0; i < N; i++) {
            CK(hipEventRecord(e0[i]));
            hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const u4 *)va[i], bytes / 16, sink);
            CK(hipEventRecord(e1[i]));
        }
        CK(hipDeviceSynchronize());
        if (p < 0) continue;
        for (int i = 0; i < N; i++) { float ms; CK(hipEventElapsedTime(&ms, e0[i], e1[i])); sum[i] += ms; mn[i] = std::min(mn[i], (double)ms); }
    }
    std::vector<int> order(N);
    for (int i = 0; i < N; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return mn[a] < mn[b]; });
    printf("%d chunks of %zu MB, %d passes; GB/s by the best pass of each chunk (mean in brackets), fastest first; index = allocation order\n", N, mb, passes);
    for (int k = 0; k < N; k++) { const int i = order[k]; printf("  chunk %3d  %7.0f  (%7.0f)\n", i, bytes / mn[i] * 1e-6, bytes / (sum[i] / passes) * 1e-6); }
    printf("fastest / slowest by best pass: %.3f\n", mn[order[N - 1]] / mn[order[0]]);
    return 0;
}
