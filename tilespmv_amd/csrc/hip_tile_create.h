// hip_tile_create.h — CSR -> Tile_matrix on the DEVICE (the device-side half; reference src/csr2tile.h:629-1020).
// A DevTile is a Tile_matrix whose member arrays live in device memory, with the uploaded CSR and the tile-ordered gather beside it.  Two consumers:
//   Tile_create_device (C ABI)            downloads it into a host Tile_matrix that is byte-identical to Tile_create's
//   tilespmv_plan_create_from_csr (C ABI) builds the plan's streams from it on the device: nothing but the CSR arrays crosses the bus
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "host_util.h"

namespace tilespmv {

struct DevTile {
    Tile_matrix T{};                       // member arrays: DEVICE pointers; counts and sizes: host values
    int rowA = 0, colA = 0;
    long long nnz = 0;                     // rowptr[rowA]
    const int *rowptr = nullptr, *colidx = nullptr;   // the uploaded CSR (device)
    const val_t *val = nullptr;
    const unsigned long long *key = nullptr;   // per nonzero, tile order: tile-row << (8 + cb_bits) | column block << 8 | local row << 4 | local column
    const int *ent = nullptr;              // per nonzero, tile order: its position in the CSR arrays
    const int *tile_bi = nullptr;          // per tile: its tile-row
    const long long *hyb_off = nullptr;    // the same as 64-bit offsets (nullptr: the matrix has no HYB tile): what plan_tile_ops.h's per-tile functions take
    const int *hyb_byte_off = nullptr;     // per tile (+ 1): first byte of a HYB tile in hybIdx (the running offset the reference calls ptroffset2, src/tilespmv_cpu.h:195-196)
    int cb_bits = 0;
    bool have_deferred = false;            // deferredcoo_* built (Tile_create_device) or skipped (plans never read them in the in-tile COO mode)
    int unsorted_rows = 0;                 // rows of the extracted matrix whose columns do not increase (the host sorts those after the download, like the reference)
    std::vector<void *> allocs;            // arrays with a hipMalloc of their own
    std::vector<void *> pools;             // pool blocks (several arrays carved from each): freed only by devtile_destroy
    double ms_upload = 0, ms_sort = 0, ms_tiles = 0, ms_select = 0, ms_pack = 0;
};

// rc 0, -1 no device, -2 int32 offsets of Tile_matrix exceeded, -3 HIP error / out of device memory
// csr_on_device: the three CSR arrays are DEVICE pointers already (row pointer based at 0; borrowed, not freed): no upload at all
int devtile_create(DevTile **out, int rowA, int colA, const MAT_PTR_TYPE *h_rowptr, const int *h_colidx, const val_t *h_val, unsigned flags, bool want_deferred, bool csr_on_device = false);
void devtile_destroy(DevTile *D);
// every member array into freshly malloc'd host arrays (Tile_destroy frees them)
int devtile_download(const DevTile *D, Tile_matrix *host);

}  // namespace tilespmv
