// host_stubs.cpp — linker stand-ins for the kernel launchers of hip_kernels.hip, for HOST-ONLY sanitizer builds of hip_plan.hip
// (scripts/asan_host.sh, scripts/tsan_host.sh).  The layout-digest build never launches anything; test scaffolding, not product code.
#include <hip/hip_runtime.h>

#include "../tilespmv_amd/csrc/hip_plan.h"

namespace tilespmv {
hipError_t launch_tiles_direct(const DevPlan &, bool, bool, bool, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_tiles_stream(const DevPlan &, const DevStream &, const DevDense &, bool, int, int, int, int, int, int, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_fallback(const DevPlan &, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_tiles_stream_mv(const DevPlan &, const DevStream &, const DevDense &, int, int, bool, int, const val_t *, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_rows_to_columns(const val_t *, int, long long, long long, val_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_columns_to_rows(const val_t *, int, long long, long long, long long, val_t *, hipStream_t) { return hipErrorNotSupported; }
int paced_team_workgroups(bool, bool, int, int) { return 192; }
hipError_t launch_pair_values(const val_t *, val_t *, const int4 *, int) { return hipErrorNotSupported; }
}
