// ThreadSanitizer driver for the multi-threaded host code (Tile_create, .mtx reader, plan layout builder): scripts/tsan_host.sh
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../tilespmv_amd/csrc/hip_plan.h"

// (the kernel launchers hip_plan.hip references come from scripts/host_stubs.cpp)

int main()
{
    std::mt19937 rng(7);
    const int rows = 48000, cols = 50003;
    std::vector<int> rp(rows + 1, 0), ci;
    for (int r = 0; r < rows; r++) {
        int k = (r % 97 == 0) ? 3000 : (int)(rng() % 40);
        int c = (int)(rng() % cols);
        for (int j = 0; j < k; j++) { ci.push_back(c); c += 1 + (int)(rng() % 7); if (c >= cols) break; }
        for (int d = -3; d <= 3; d++) if (r + d >= 0 && r + d < cols && (r + d) > ci.back()) ci.push_back(r + d);
        rp[r + 1] = (int)ci.size();
    }
    std::vector<double> v(ci.size());
    for (size_t i = 0; i < v.size(); i++) v[i] = (double)(i % 10);
    for (unsigned flags = 2; flags <= 3; flags++) {
        Tile_matrix T;
        Tile_create_ex(&T, rows, cols, (int)ci.size(), rp.data(), ci.data(), v.data(), flags);
        printf("flags %u tiles %d\n", flags, T.tilenum);
        // two host threads build differently tuned plan layouts of the same matrix at the same time (the knobs travel in
        // tilespmv_plan_options, nothing in the environment): digests must equal the ones of the serial builds
        unsigned long long serial[4], par[4];
        auto build = [&](int i, unsigned long long *out) {
            tilespmv_plan_options o;
            tilespmv_plan_options_init(&o);
            o.entry_mode = i % 3; o.strip_cost = 200 + 300 * i; o.entry_ordered = i & 1; o.wg_strips = (i & 2) ? 32 : 16;
            if (i == 3) o.coo_mode = TILESPMV_COO_FALLBACK;
            if (i == 2) o.x_window = 1;
            if (tilespmv_plan_layout_digest(&T, rows, cols, (int)ci.size(), &o, out + i, nullptr) != 0) out[i] = 0;
        };
        for (int i = 0; i < 4; i++) build(i, serial);
        std::thread a([&] { build(0, par); build(2, par); }), b([&] { build(1, par); build(3, par); });
        a.join(); b.join();
        int same = 0;
        for (int i = 0; i < 4; i++) same += serial[i] != 0 && serial[i] == par[i];
        printf("flags %u plan layouts from two threads: %d of 4 equal the serial digests\n", flags, same);
        Tile_destroy(&T);
    }
    const char *path = "/tmp/tilespmv_tsan.mtx";
    FILE *f = fopen(path, "w");
    fprintf(f, "%%%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n", cols, cols, 400000);
    for (int i = 0; i < 400000; i++) { int a = 1 + (int)(rng() % cols), b = 1 + (int)(rng() % cols); fprintf(f, "%d %d %.6e\n", a > b ? a : b, a > b ? b : a, (double)(rng() % 1000) / 7.0); }
    fclose(f);
    int m, n, nnz, sym; int *prp, *pci; double *pv;
    int rc = mmio_allinone(&m, &n, &nnz, &sym, &prp, &pci, &pv, (char *)path);
    printf("mmio rc %d nnz %d\n", rc, nnz);
    if (rc == 0) { free(prp); free(pci); free(pv); }
    return 0;
}
