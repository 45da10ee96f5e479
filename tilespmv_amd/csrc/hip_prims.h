// hip_prims.h — the device-wide primitives the on-device builders use (stable radix sorts, exclusive scans, run-length encoding), instantiated ONCE.
// rocPRIM's radix sort alone compiles to ~2.5 MB of gfx950 code per key / value type pair; rounds 4-5 instantiated the same (u64 key, int value) sort in two translation units
// (hip_tile_create.hip, hip_plan_device.hip) — 5.3 MB of each 11-MB library.  The wrappers keep rocPRIM's two-call convention: tmp == nullptr asks for tmp_b.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

namespace tilespmv {
namespace prims {

typedef unsigned long long u64;

// out[i] = in[0] + ... + in[i - 1] (int, init 0); in == out allowed
hipError_t scan_int(void *tmp, size_t &tmp_b, const int *in, int *out, size_t n, hipStream_t st);

// Stable LSD radix sorts over bits [b0, b1) of the key.  Double buffers as pointer pairs: on return *_cur holds the sorted data, *_alt the scratch copy (they may have swapped).
hipError_t sort_pairs_u64_int(void *tmp, size_t &tmp_b, u64 *&k_cur, u64 *&k_alt, int *&v_cur, int *&v_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st);
hipError_t sort_pairs_u32_int(void *tmp, size_t &tmp_b, unsigned *&k_cur, unsigned *&k_alt, int *&v_cur, int *&v_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st);
hipError_t sort_keys_u64(void *tmp, size_t &tmp_b, u64 *&k_cur, u64 *&k_alt, size_t n, unsigned b0, unsigned b1, hipStream_t st);

// runs of equal values of (in[i] >> shift): uniq[r] = the value, counts[r] = the run's length, *nruns = the number of runs
hipError_t rle_u64(void *tmp, size_t &tmp_b, const u64 *in, unsigned shift, unsigned n, u64 *uniq, int *counts, int *nruns, hipStream_t st);

}  // namespace prims
}  // namespace tilespmv
