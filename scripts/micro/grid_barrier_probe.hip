// grid_barrier_probe.hip — what would ONE grid-wide barrier cost inside a persistent "K dependent SpMVs per launch" kernel,
// next to the kernel boundary it would replace?  (keep the matrix in the per-XCD L2s across
// iterations of a solver.)  A dependent SpMV chain needs, between two iterations, every workgroup's y stores visible to every
// other workgroup's x gathers: a grid barrier with an agent-scope release before and an acquire after it.
// Measured here, per iteration, for G resident workgroups of 256 threads:
//   launches      K back-to-back launches of a kernel that does the same token work (the boundary this chip charges)
//   flat barrier  one monotonic counter: lane 0 release fence -> agent atomic add -> relaxed sc1 poll (+ s_sleep) -> acquire fence
//   xcd barrier   per-XCC counters on lines of their own, the last arriver of each XCC goes to a top counter and then
//                 opens its XCC's generation word (MI355X_MICROARCH.md "barrier-xcd")
// Every spin is bounded (a barrier that cannot complete sets a flag and the kernel ends): the probe cannot hang the GPU.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier_probe.hip -o scripts/micro/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar {
    unsigned top; unsigned pad0[31];
    unsigned xcc_count[8][32];   // one 128-B line per XCC
    unsigned xcc_gen[8][32];
    unsigned xcc_size[8][32];    // workgroups resident on each XCC (census of the first phase)
    unsigned nxcc; unsigned failed; unsigned pad1[30];
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID, bits 0-3

__device__ bool spin_until(const unsigned *p, unsigned target, unsigned *failed)
{
    for (int i = 0; i < (1 << 22); i++) {
        if ((int)(ld_sc1(p) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    *failed = 1u;
    return false;
}

// token work of one "iteration": every workgroup writes 2 KB and reads 2 KB somebody else wrote in the previous iteration
__device__ __forceinline__ void token_work(double *buf, int it, int G)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const double *src = buf + (size_t)((it & 1) * G + (b * 37 + 11) % G) * 256;
    double *dst = buf + (size_t)(((it + 1) & 1) * G + b) * 256;
    dst[t] = src[t] + 1.0;
}

__global__ __launch_bounds__(256) void k_one(double *buf, int it, int G) { token_work(buf, it, G); }

template <int MODE>   // 0 flat, 1 xcd-hierarchical
__global__ __launch_bounds__(256) void k_persistent(double *buf, Bar *B, int K, int G)
{
    __shared__ unsigned s_go;
    const unsigned x = xcc_id();
    unsigned my_size = 0, nx = 0;
    if (MODE == 1) {   // census: how many workgroups sit on my XCC, how many XCCs are populated (one flat barrier)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&B->xcc_size[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&B->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(&B->top, (unsigned)G, &B->failed);
            my_size = ld_sc1(&B->xcc_size[x][0]);
            for (int q = 0; q < 8; q++) nx += ld_sc1(&B->xcc_size[q][0]) > 0;
        }
    }
    const unsigned base = MODE == 1 ? (unsigned)G : 0u;   // the census used `top` once
    for (int it = 0; it < K; it++) {
        token_work(buf, it, G);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&B->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(&B->top, (unsigned)G * (unsigned)(it + 1), &B->failed);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned prev = __hip_atomic_fetch_add(&B->xcc_count[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1u == my_size * (unsigned)(it + 1)) {   // last arriver of this XCC: up to the top counter, then open the XCC
                    __hip_atomic_fetch_add(&B->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    spin_until(&B->top, base + nx * (unsigned)(it + 1), &B->failed);
                    __hip_atomic_store(&B->xcc_gen[x][0], (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else spin_until(&B->xcc_gen[x][0], (unsigned)(it + 1), &B->failed);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_go = 1u;
        }
        __syncthreads();
        if (ld_sc1(&B->failed)) return;
    }
}

int main()
{
    const int K = 400;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%6s %22s %22s %22s\n", "WGs", "us/iteration launches", "flat barrier", "xcd barrier");
    for (int G : {256, 512, 768, 1024}) {
        double *buf; Bar *B;
        CK(hipMalloc(&buf, (size_t)2 * G * 256 * sizeof(double))); CK(hipMemset(buf, 0, (size_t)2 * G * 256 * sizeof(double)));
        CK(hipMalloc(&B, sizeof(Bar)));
        float ms[3] = {0, 0, 0};
        unsigned failed[3] = {0, 0, 0};
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, 0, buf, i, G);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, 0, buf, i, G);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[0], a, b));
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 2; rep++) {   // first repetition warms up
                CK(hipMemset(B, 0, sizeof(Bar)));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(a, 0));
                if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(G), dim3(256), 0, 0, buf, B, K, G);
                else hipLaunchKernelGGL(k_persistent<1>, dim3(G), dim3(256), 0, 0, buf, B, K, G);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[1 + mode], a, b));
                Bar h; CK(hipMemcpy(&h, B, sizeof(Bar), hipMemcpyDeviceToHost));
                failed[1 + mode] = h.failed;
            }
        }
        std::vector<double> hb((size_t)2 * G * 256);
        CK(hipMemcpy(hb.data(), buf, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
        printf("%6d %22.3f %18.3f%4s %18.3f%4s\n", G, ms[0] * 1000.0 / K, ms[1] * 1000.0 / K, failed[1] ? " TO" : "", ms[2] * 1000.0 / K, failed[2] ? " TO" : "");
        CK(hipFree(buf)); CK(hipFree(B));
    }
    printf("(TO = a bounded spin ran out: not every workgroup was resident; the figure is then meaningless)\n");
    return 0;
}
