// Would pinning column slices of x to XCDs pay?  (round 4, after the column panels: a panel pass makes all eight XCDs sweep the SAME slice of x, one kernel per slice, and pays a
// read-modify-write of y per pass.  The other way round: workgroup b runs on XCD b & 7 (round-robin dispatch), so give XCD k only the entries whose columns lie in slice k of x —
// the slice stays in that XCD's 4-MB L2 for the whole launch, no pass boundaries — and combine the eight partial row sums per row.)
// A model of the entry phase only: n rows, d scattered entries per row, 16-byte records streamed once, groups of 1,536 rows with an LDS slab, 256 threads per workgroup.
//   variant 0  today's unpanelled form: one workgroup per group, columns anywhere in x, plain store of the slab
//   variant 1  eight workgroups per group, workgroup (g, k) takes the group's entries of slice k; slab -> y by coalesced atomic adds (every row)
//   variant 2  as 1, the slab stored to a partial vector part[k][row] (nontemporal); a second kernel sums the eight partials into y
//   variant 3  as 1 without any output (the gather rate alone)
//   variant 4  as 1, atomic adds only for rows the workgroup touched (sum != 0)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/micro/xcd_columns.hip -o scripts/micro/xcd_columns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int GROUP_ROWS = 1536;
struct Rec { double v; unsigned col; unsigned row; };   // 16 bytes

__device__ inline unsigned long long mix(unsigned long long s) { s ^= s >> 33; s *= 0xff51afd7ed558ccdull; s ^= s >> 33; s *= 0xc4ceb9fe1a85ec53ull; s ^= s >> 33; return s; }

// records of group g: per_wg records per workgroup, workgroup-major.  sliced: workgroup (g, k)'s columns in slice k; else anywhere
__global__ void k_fill(Rec *rec, long long total, int per_wg, unsigned n, int sliced)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long h = mix((unsigned long long)i * 0x9E3779B97F4A7C15ull + 1);
        const long long wg = i / per_wg;
        unsigned col;
        if (sliced) { const unsigned k = (unsigned)(wg & 7), sl = n / 8; col = k * sl + (unsigned)((h >> 8) % sl); }
        else col = (unsigned)((h >> 8) % n);
        Rec r; r.v = 1.0; r.col = col; r.row = (unsigned)((h >> 40) % GROUP_ROWS);
        rec[i] = r;
    }
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_entries(const Rec *__restrict__ rec, int per_wg, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ part, long long n)
{
    __shared__ double slab[GROUP_ROWS];
    const int tid = threadIdx.x;
    const long long bid = blockIdx.x;
    const long long g = VARIANT == 0 ? bid : (bid >> 3);
    const int k = (int)(bid & 7);
    for (int i = tid; i < GROUP_ROWS; i += 256) slab[i] = 0;
    __syncthreads();
    const Rec *mine = rec + bid * per_wg;
    for (int e0 = 0; e0 < per_wg; e0 += 256 * 6) {
        Rec r[6]; double xv[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const int e = e0 + q * 256 + tid;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(mine + min(e, per_wg - 1)));
            r[q].v = __hiloint2double((int)raw.y, (int)raw.x); r[q].col = raw.z; r[q].row = raw.w;
        }
#pragma unroll
        for (int q = 0; q < 6; q++) xv[q] = x[r[q].col];
#pragma unroll
        for (int q = 0; q < 6; q++) if (e0 + q * 256 + tid < per_wg) atomicAdd(&slab[r[q].row], r[q].v * xv[q]);
    }
    __syncthreads();
    double *yg = y + g * GROUP_ROWS;
    if (VARIANT == 0) { for (int i = tid; i < GROUP_ROWS; i += 256) __builtin_nontemporal_store(slab[i], &yg[i]); }
    else if (VARIANT == 1) { for (int i = tid; i < GROUP_ROWS; i += 256) atomicAdd(&yg[i], slab[i]); }
    else if (VARIANT == 2) { double *pg = part + (long long)k * n + g * GROUP_ROWS; for (int i = tid; i < GROUP_ROWS; i += 256) __builtin_nontemporal_store(slab[i], &pg[i]); }
    else if (VARIANT == 4) { for (int i = tid; i < GROUP_ROWS; i += 256) { const double s = slab[i]; if (s != 0) atomicAdd(&yg[i], s); } }
    else { if (slab[tid] == 1.2345e300) yg[0] = 1; }
}

__global__ __launch_bounds__(256) void k_sum8(const double *__restrict__ part, double *__restrict__ y, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += __builtin_nontemporal_load(&part[(long long)k * n + i]);
    __builtin_nontemporal_store(s, &y[i]);
}

template <int VARIANT>
static double run(const Rec *rec, int per_wg, long long wgs, const double *x, double *y, double *part, long long n)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 10;
    for (int it = -2; it < reps; it++) {
        if (it == 0) CK(hipEventRecord(a));
        if (VARIANT == 1 || VARIANT == 4) CK(hipMemsetAsync(y, 0, n * 8));   // stands for the unit kernel's store of y
        hipLaunchKernelGGL(k_entries<VARIANT>, dim3((unsigned)wgs), dim3(256), 0, 0, rec, per_wg, x, y, part, n);
        if (VARIANT == 2) hipLaunchKernelGGL(k_sum8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, part, y, n);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const long long rows_m[] = {2, 4, 4, 8};
    const int deg[] = {3, 8, 3, 8};
    for (int c = 0; c < 4; c++) {
        const long long groups = rows_m[c] * 1000000 / GROUP_ROWS, n = groups * GROUP_ROWS;
        const int d = deg[c];
        const long long total = n * d;
        Rec *rec; double *x, *y, *part;
        CK(hipMalloc(&rec, total * sizeof(Rec))); CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&part, n * 8 * 8));
        CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
        printf("rows %lld (x %lld MB, %lld MB per slice), %d scattered entries per row, %lld M entries, records %lld MB\n", n, n * 8 >> 20, n >> 20, d, total / 1000000, total * 16 >> 20);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, rec, total, GROUP_ROWS * d, (unsigned)n, 0);
        const double t0 = run<0>(rec, GROUP_ROWS * d, groups, x, y, part, n);
        printf("  0 one workgroup per group, columns anywhere, store     %.4f ms  %.0f G entries/s\n", t0, total / t0 * 1e-6);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, rec, total, GROUP_ROWS * d / 8, (unsigned)n, 1);
        const double t3 = run<3>(rec, GROUP_ROWS * d / 8, groups * 8, x, y, part, n);
        printf("  3 slice k on XCD k, no output                          %.4f ms  %.0f G entries/s\n", t3, total / t3 * 1e-6);
        const double t1 = run<1>(rec, GROUP_ROWS * d / 8, groups * 8, x, y, part, n);
        printf("  1 slice k on XCD k, atomic add of every row (+ memset)  %.4f ms  %.0f G entries/s\n", t1, total / t1 * 1e-6);
        const double t4 = run<4>(rec, GROUP_ROWS * d / 8, groups * 8, x, y, part, n);
        printf("  4 slice k on XCD k, atomic add of touched rows (+ memset) %.4f ms  %.0f G entries/s\n", t4, total / t4 * 1e-6);
        const double t2 = run<2>(rec, GROUP_ROWS * d / 8, groups * 8, x, y, part, n);
        printf("  2 slice k on XCD k, eight partial vectors + sum kernel %.4f ms  %.0f G entries/s\n", t2, total / t2 * 1e-6);
        CK(hipFree(rec)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(part));
    }
    return 0;
}
