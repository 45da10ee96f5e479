// host_stubs.cpp — linker stand-ins for the kernel launchers of hip_kernels.hip, for HOST-ONLY sanitizer builds of hip_plan.hip
// (scripts/asan_host.sh, scripts/tsan_host.sh).  The layout-digest build never launches anything; test scaffolding, not product code.
#include <hip/hip_runtime.h>

#include "../tilespmv_amd/csrc/hip_plan.h"

#include "../tilespmv_amd/csrc/hip_plan_device.h"

// (every stand-in returns "not supported": the host-only builds never reach a device path — tilespmv_plan_layout_digest builds from a host Tile_matrix and launches nothing)
namespace tilespmv {
hipError_t launch_tiles_direct(const DevPlan &, bool, bool, bool, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_tiles_stream(const DevPlan &, const DevStream &, const DevDense &, bool, int, int, int, int, int, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_fallback(const DevPlan &, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_tiles_stream_mv(const DevPlan &, const DevStream &, const DevDense &, int, int, bool, int, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_rows_to_columns(const val_t *, int, long long, long long, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_columns_to_rows(const val_t *, int, long long, long long, long long, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_pair_values(const val_t *, val_t *, const int4 *, int) { return hipErrorNotSupported; }
hipError_t launch_permute_vector(const val_t *, val_t *, const int *, long long, int, hipStream_t) { return hipErrorNotSupported; }
// the device-side builders (hip_tile_create.hip, hip_plan_device.hip)
int devtile_create(DevTile **, int, int, const MAT_PTR_TYPE *, const int *, const val_t *, unsigned, bool, bool) { return -1; }
void devtile_destroy(DevTile *) {}
int devtile_download(const DevTile *, Tile_matrix *) { return -3; }
void DevCounts::release() {}
void DevLists::release() {}
int dev_fetch_ints(const int *, const long long *, int, int *) { return -3; }
int dev_count(const DevShard &, int, DevCounts *, hvec<int> &, long long *, long long *) { return -3; }
int dev_pattern_sample(const DevShard &, int, int, int, std::vector<unsigned long long> &) { return -3; }
int dev_emit(const DevShard &, const DevCounts &, const hvec<long long> &, const hvec<long long> &, const hvec<long long> &, const std::vector<unsigned char> &, const std::vector<unsigned char> &, long long, const EmitOut &) { return -3; }
int dev_fetch_word0(const uint4 *, long long, hvec<unsigned> &) { return -3; }
int dev_pack_desc(const uint4 *, const uint2 *, const uint4 *, const int4 *, int, UDesc *, URow *, uint4 *) { return -3; }
int dev_shift_histogram(const UDesc *, long long, unsigned long long *) { return -3; }
int dev_dict_patterns(const UDesc *, long long, size_t, std::vector<uint4> &, DictRanges *, bool *) { return -3; }
int dev_compact_desc(const UDesc *, long long, const uint4 *, DictRanges, int, unsigned *) { return -3; }
int dev_pool_dict(const UDesc *, const URow *, long long, size_t, std::vector<uint4> &, bool *) { return -3; }
int dev_pool_compact(const UDesc *, const URow *, long long, const uint4 *, int, int, void *) { return -3; }
int dev_entry_lists(const val_t *, const int *, const unsigned char *, long long, const std::vector<STask> &, int, int, int, bool, int, int, DevLists *) { return -3; }
}
