// Is memory from hipMemCreate / hipMemMap safe to use the way the plan's arena uses hipMalloc memory?  (round 4: plans whose blocks were virtual ranges over separately created
// physical chunks gave wrong results on SMALL plans and a GPU fault; large plans were fine.)  This probe never dereferences anything read from the memory under test: it only
// copies patterns in (small and large hipMemcpy, hipMemset), reads them back through hipMemcpy and through a bounds-safe kernel, frees, and repeats — so that a stale
// translation or a copy path that misses shows up as a mismatch count, not as a fault.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/vmm_probe.hip -o scripts/micro/vmm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_read(const unsigned *__restrict__ src, unsigned *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct Range { void *ptr; size_t size; std::vector<hipMemGenericAllocationHandle_t> h; };

static Range vmm_alloc(size_t bytes, size_t chunk)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t total = (bytes + gran - 1) / gran * gran;
    Range R{nullptr, total, {}};
    CK(hipMemAddressReserve(&R.ptr, total, 0, nullptr, 0));
    for (size_t off = 0; off < total; off += chunk) {
        const size_t sz = std::min(chunk, total - off);
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, sz, &prop, 0));
        CK(hipMemMap((char *)R.ptr + off, sz, 0, h, 0));
        R.h.push_back(h);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(R.ptr, total, &acc, 1));
    return R;
}

static void vmm_free(Range &R, bool free_va)
{
    CK(hipMemUnmap(R.ptr, R.size));
    for (auto h : R.h) CK(hipMemRelease(h));
    if (free_va) CK(hipMemAddressFree(R.ptr, R.size));
}

int main(int argc, char **argv)
{
    const bool free_va = argc < 2 || atoi(argv[1]) != 0;
    const bool with_memset = argc < 3 || atoi(argv[2]) != 0;
    size_t gran = 0;
    { hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)); }
    printf("granularity %zu KB; VA ranges %s after use; %s\n", gran >> 10, free_va ? "freed" : "kept reserved", with_memset ? "hipMemset before the copies" : "no memset");
    const size_t bytes = (size_t)3 << 20, nw = bytes / 4;
    unsigned *out; CK(hipMalloc(&out, bytes));
    std::vector<unsigned> host(nw), back(nw), back2(nw);
    long long bad_copy = 0, bad_kernel = 0, bad_zero = 0;
    void *last = nullptr; int same_va = 0;
    for (int it = 0; it < 60; it++) {
        Range R = vmm_alloc(bytes, (size_t)1 << 20);
        same_va += R.ptr == last; last = R.ptr;
        if (with_memset) CK(hipMemset(R.ptr, 0, R.size));
        // many small uploads at odd offsets (what a plan's upload() does), leaving gaps that must read as zero when memset
        for (size_t i = 0; i < nw; i++) host[i] = 0;
        for (int s = 0; s < 40; s++) {
            const size_t off = ((size_t)s * 70001 + it * 131) % (nw - 5000), len = 16 + (s * 977 + it * 13) % 4000;
            for (size_t i = 0; i < len; i++) host[off + i] = (unsigned)(it * 1000003u + s * 7919u + i);
            CK(hipMemcpy((unsigned *)R.ptr + off, &host[off], len * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(back.data(), R.ptr, bytes, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, 0, (const unsigned *)R.ptr, out, nw);
        CK(hipMemcpy(back2.data(), out, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nw; i++) {
            const bool written = host[i] != 0;
            if (written) { bad_copy += back[i] != host[i]; bad_kernel += back2[i] != host[i]; }
            else if (with_memset) bad_zero += (back[i] != 0) + (back2[i] != 0);
        }
        vmm_free(R, free_va);
        if (it % 3 == 0) { void *p; CK(hipMalloc(&p, (size_t)(1 + it % 5) << 20)); CK(hipMemset(p, 0xAB, (size_t)1 << 20)); CK(hipDeviceSynchronize()); CK(hipFree(p)); }
    }
    printf("60 rounds: words wrong after copy-back %lld, through a kernel %lld, gaps not zero %lld; the same VA came back %d times\n", bad_copy, bad_kernel, bad_zero, same_va);
    return 0;
}
