// host_util.h — small host-side helpers shared by the product's C++ sources.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/tilespmv.h"

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
    return p;
}

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

inline int tile_rowlen(int bi, int tilem, int rowA) { return bi == tilem - 1 ? rowA - (tilem - 1) * BS : BS; }
inline int tile_collen(int cb, int tilen, int colA) { return cb == tilen - 1 ? colA - (tilen - 1) * BS : BS; }

// Row-block schedule of the reference (src/tilespmv_cpu.h:68-118); arrays are malloc'd.
int build_rowblock_schedule(const Tile_matrix *T, unsigned int **rowidx, int **colstart, int **colstop);

}  // namespace tilespmv
