// launch_floor.hip — what does a short dependent kernel cost on this chip, back to back on one stream?
// The cache-resident BASELINE configs (scircuit ~15 MB, webbase-1M ~57 MB) have roofline times of 2-8 us, so
// their SpMV time is set by (a) the kernel-to-kernel turnaround and (b) the depth of the dependent load chain
// inside one strip (task -> entries -> x gather -> y store).  This probe times both, so that DESIGN.md can say
// what the floor is for a given grid size and chain depth.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_floor.hip -o scripts/micro/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// DEPTH dependent loads per lane (pointer chase through idx[]), then one store
template <int DEPTH>
__global__ __launch_bounds__(256) void k_chain(const int *__restrict__ idx, double *__restrict__ y, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int j = i;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) j = idx[j];
    y[i] = (double)j;
}

int main()
{
    const int NMAX = 8192 * 256;
    std::vector<int> h(NMAX);
    // a permutation with stride so that consecutive hops touch different lines, all within NMAX
    for (int i = 0; i < NMAX; i++) h[i] = (int)(((long long)i * 1021 + 17) % NMAX);
    int *d_idx; double *d_y;
    CK(hipMalloc(&d_idx, NMAX * sizeof(int))); CK(hipMalloc(&d_y, NMAX * sizeof(double)));
    CK(hipMemcpy(d_idx, h.data(), NMAX * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 2000;
    printf("%8s %6s %10s\n", "blocks", "depth", "us/launch");
    for (int blocks : {64, 256, 700, 1024, 2048, 4096, 8192}) {
        for (int depth = 0; depth <= 4; depth++) {
            auto launch = [&]() {
                const int n = blocks * 256;
                switch (depth) {
                case 0: hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, d_idx, d_y, n); break;
                case 1: hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, d_idx, d_y, n); break;
                case 2: hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(256), 0, 0, d_idx, d_y, n); break;
                case 3: hipLaunchKernelGGL(k_chain<3>, dim3(blocks), dim3(256), 0, 0, d_idx, d_y, n); break;
                default: hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(256), 0, 0, d_idx, d_y, n); break;
                }
            };
            for (int i = 0; i < 200; i++) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < reps; i++) launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            printf("%8d %6d %10.3f\n", blocks, depth, ms * 1000.0 / reps);
        }
    }
    return 0;
}
