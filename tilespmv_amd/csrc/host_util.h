// host_util.h — small host-side helpers shared by the product's C++ sources.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "../../include/tilespmv.h"

// functions shared by host code and device kernels (the .cpp sources are compiled as plain C++: there the qualifiers vanish)
#if defined(__HIP__)
#define TILESPMV_HD __host__ __device__
#else
#define TILESPMV_HD
#endif

namespace tilespmv {

typedef MAT_VAL_TYPE val_t;
constexpr int BS = TILESPMV_BLOCK_SIZE;

inline int host_threads()
{
    static int n = [] {
        const char *e = getenv("TILESPMV_NUM_THREADS");
        if (!e) e = getenv("OMP_NUM_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) v = (int)std::thread::hardware_concurrency();
        if (!e) v = std::min(v, 32);  // default: enough to saturate host memory bandwidth, polite on shared nodes
        return v > 0 ? std::min(v, 256) : 1;
    }();
    return n;
}

// Dynamic-chunk parallel loop: body(begin, end, thread_id).  Serial when the range is small.
template <class F>
inline void parallel_chunks(int64_t n, int64_t chunk, F body)
{
    int nt = host_threads();
    if (n <= chunk || nt == 1) { if (n > 0) body((int64_t)0, n, 0); return; }
    nt = (int)std::min<int64_t>(nt, (n + chunk - 1) / chunk);
    std::atomic<int64_t> next(0);
    std::vector<std::thread> pool;
    auto worker = [&](int tid) {
        for (;;) {
            int64_t b = next.fetch_add(chunk);
            if (b >= n) break;
            body(b, std::min(n, b + chunk), tid);
        }
    };
    for (int t = 1; t < nt; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool) th.join();
}

template <class T>
inline T *zalloc(size_t n)
{
    T *p = (T *)calloc(n ? n : 1, sizeof(T));
    if (!p) { fprintf(stderr, "tilespmv: out of host memory (%zu x %zu bytes)\n", n, sizeof(T)); exit(2); }
    // Large arrays: ask for transparent huge pages (round 5; what numpy does for its own large arrays).  A calloc of this size is a fresh anonymous mapping that is zeroed page
    // by page at first touch: the per-tile arrays of config 4's Tile_matrix (1.5 GB) and the plan builder's staging arrays cost 4-KB page faults by the hundred thousand — on the
    // GPU box's host 2/3 of Tile_create's time and 1/3 of plan creation's (config 4: 0.26 -> 0.08-0.14 s and 0.31 -> 0.19 s, scripts/archive/rounds/r5_hugepages.sh); with THP in
    // "madvise" mode (this image) the advice turns them into 2-MB faults.  Still plain malloc memory: the caller frees it with free() as the reference's API requires.
    // TILESPMV_HUGEPAGES=0 switches the advice off (a host whose memory is too fragmented to have huge pages at hand compacts synchronously inside such a fault).
    static const bool thp = [] { const char *e = getenv("TILESPMV_HUGEPAGES"); return !(e && *e && atoi(e) == 0); }();
    const size_t bytes = n * sizeof(T);
    if (thp && bytes >= ((size_t)8 << 20)) {
        const uintptr_t lo = ((uintptr_t)p + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), hi = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    return p;
}

// std::vector with the same advice for its storage: the plan builder's per-tile-row arrays (a million tile-rows: 40-50 MB each for the counts, 8 MB per prefix array) were first touched
// through 4-KB page faults — 16 of the 20 ms of COUNT in the device-built plan of config 4.  hvec<T> is a std::vector whose allocations of 8 MB or more carry MADV_HUGEPAGE.
inline void advise_huge(void *p, size_t bytes)
{
    static const bool thp = [] { const char *e = getenv("TILESPMV_HUGEPAGES"); return !(e && *e && atoi(e) == 0); }();
    if (!thp || bytes < ((size_t)8 << 20)) return;
    const uintptr_t lo = ((uintptr_t)p + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1), hi = ((uintptr_t)p + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
}
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n)
    {
        T *p = (T *)malloc((n ? n : 1) * sizeof(T));
        if (!p) throw std::bad_alloc();
        advise_huge(p, n * sizeof(T));
        return p;
    }
    void deallocate(T *p, size_t) { free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using hvec = std::vector<T, HugeAlloc<T>>;

// In-place exclusive prefix sum with 64-bit accumulation; aborts if a prefix leaves int range
// (the reference's int products overflow silently: SURVEY.md S6).
inline void exclusive_scan_checked(int *a, int64_t n, const char *what)
{
    int64_t run = 0;
    for (int64_t i = 0; i < n; i++) {
        int v = a[i];
        if (run > INT32_MAX) { fprintf(stderr, "tilespmv: %s exceeds the int32 offsets of Tile_matrix\n", what); exit(2); }
        a[i] = (int)run;
        run += v;
    }
}

// The same scan for K arrays of n ints at once, in parallel (round 5: Tile_create ran its 13 per-tile scans as 13 serial loops on 13 threads — 140 ms of config 4's 570 ms on
// 8 cores — and the scan of deferredcoo_ptr as one serial loop): pass A sums every array over chunks, a serial prefix over the chunk sums, pass B scans every chunk in place from
// its offset.  Same result, same abort on a prefix that leaves the int range.
inline void exclusive_scan_checked_multi(int *const *arrays, const char *const *names, int K, int64_t n)
{
    const int64_t chunk = std::max<int64_t>(1 << 14, (n + 4LL * host_threads() - 1) / (4LL * host_threads()));
    const int64_t nch = (n + chunk - 1) / chunk;
    std::vector<int64_t> sums((size_t)(nch * K), 0);
    parallel_chunks(nch, 1, [&](int64_t b, int64_t e, int) {
        for (int64_t c = b; c < e; c++)
            for (int k = 0; k < K; k++) {
                const int *a = arrays[k];
                int64_t run = 0;
                for (int64_t i = c * chunk; i < std::min(n, (c + 1) * chunk); i++) run += a[i];
                sums[(size_t)(c * K + k)] = run;
            }
    });
    for (int k = 0; k < K; k++) {
        int64_t run = 0;
        for (int64_t c = 0; c < nch; c++) { const int64_t v = sums[(size_t)(c * K + k)]; sums[(size_t)(c * K + k)] = run; run += v; }
    }
    std::atomic<int> overflow(-1);
    parallel_chunks(nch, 1, [&](int64_t b, int64_t e, int) {
        for (int64_t c = b; c < e; c++)
            for (int k = 0; k < K; k++) {
                int *a = arrays[k];
                int64_t run = sums[(size_t)(c * K + k)];
                for (int64_t i = c * chunk; i < std::min(n, (c + 1) * chunk); i++) {
                    const int v = a[i];
                    if (run > INT32_MAX) overflow.store(k);
                    a[i] = (int)run;
                    run += v;
                }
            }
    });
    if (overflow.load() >= 0) { fprintf(stderr, "tilespmv: %s exceeds the int32 offsets of Tile_matrix\n", names[overflow.load()]); exit(2); }
}

// free() of large temporaries on a detached thread (munmap of hundreds of MB takes tens of milliseconds)
inline void free_later(std::vector<void *> ptrs)
{
    ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), (void *)nullptr), ptrs.end());
    if (ptrs.empty()) return;
    std::thread([ptrs]() { for (void *q : ptrs) free(q); }).detach();
}

TILESPMV_HD inline int tile_rowlen(int bi, int tilem, int rowA) { return bi == tilem - 1 ? rowA - (tilem - 1) * BS : BS; }
TILESPMV_HD inline int tile_collen(int cb, int tilen, int colA) { return cb == tilen - 1 ? colA - (tilen - 1) * BS : BS; }

// First-element-pivot partition sort of the reference (host_tile_create.cpp; also used by the device builder's download for rows of the extracted matrix that arrive unsorted)
void pivot_sort(int *key, val_t *val, int n);

// Row-block schedule of the reference (src/tilespmv_cpu.h:68-118); arrays are malloc'd.
int build_rowblock_schedule(const Tile_matrix *T, unsigned int **rowidx, int **colstart, int **colstop);

}  // namespace tilespmv
