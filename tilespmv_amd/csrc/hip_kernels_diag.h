// hip_kernels_diag.h — TIMING-ONLY diagnostics of hip_kernels.hip.  Included only with -DTILESPMV_DIAG (make VARIANT=_name EXTRA_DEFS=...: the product libraries
// never contain any of this).  Most of these variants return WRONG rows by construction — they exist to price one part of a kernel (scripts/ablate_entries.sh,
// scripts/archive/rounds/r4_slices*.sh, scripts/archive/rounds/r5_pool_abl.sh, scripts/stamps_probe.py, scripts/archive/rounds/r4_gather_policy.sh).  Every macro names the locals of the function it is used in.
#pragma once

// ---- cache policy of the scattered x gathers of the workgroup entry phase: TILESPMV_GATHER_POLICY 0 default, 1 nontemporal, 2 agent scope (sc1), 3 system scope (sc0 sc1)
#if defined(TILESPMV_GATHER_POLICY) && TILESPMV_GATHER_POLICY == 1
#define TSPMV_DIAG_GATHER_X(p) __builtin_nontemporal_load(p)
#elif defined(TILESPMV_GATHER_POLICY) && TILESPMV_GATHER_POLICY == 2
#define TSPMV_DIAG_GATHER_X(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#elif defined(TILESPMV_GATHER_POLICY) && TILESPMV_GATHER_POLICY == 3
#define TSPMV_DIAG_GATHER_X(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#else
#define TSPMV_DIAG_GATHER_X(p) (*(p))
#endif

// ---- wg_entry_trips: TILESPMV_ABL 1 no LDS adds, 2 contiguous instead of gathered x, 5 one extra 2-byte stream load per entry
#ifdef TILESPMV_ABL
#define TSPMV_DIAG_TRIP_DECL val_t abl_acc = 0;
#else
#define TSPMV_DIAG_TRIP_DECL
#endif
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 5
#define TSPMV_DIAG_TRIP_RECORD(r, q, e0) r[q].w += reinterpret_cast<const unsigned short *>(base)[min(e0 + NT * q + tid, ge - 1)] & 1u;
#else
#define TSPMV_DIAG_TRIP_RECORD(r, q, e0)
#endif
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 2
#define TSPMV_DIAG_TRIP_GATHERS _Pragma("unroll") for (int q = 0; q < CT; q++) xx[q] = x[(e0 + NT * q + tid) & 0xFFFFF];
#else
#define TSPMV_DIAG_TRIP_GATHERS
#endif
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 1
#define TSPMV_DIAG_TRIP_ADDS_REPLACED 1
#define TSPMV_DIAG_TRIP_ADDS _Pragma("unroll") for (int q = 0; q < CT; q++) abl_acc += erec_val(rr[q]) * xx[q] + (val_t)(rr[q].w & dmask); \
        if (e0 + NT * CT >= ge) atomicAdd(&sy[tid], (lacc_t)abl_acc);
#else
#define TSPMV_DIAG_TRIP_ADDS_REPLACED 0
#define TSPMV_DIAG_TRIP_ADDS
#endif

// ---- k_units: extra LDS per workgroup (what fewer resident workgroups cost); TILESPMV_ABL 7: a unit whose column block equals the previous unit's skips its gather
#ifdef TILESPMV_ABL_LDS_PAD
#define TSPMV_DIAG_UNITS_LDS_PAD __shared__ unsigned s_pad[TILESPMV_ABL_LDS_PAD / 4]; \
    if (rowA < 0) s_pad[threadIdx.x] = 1u, y[0] = (val_t)s_pad[(threadIdx.x * 7) % (TILESPMV_ABL_LDS_PAD / 4)];
#else
#define TSPMV_DIAG_UNITS_LDS_PAD
#endif
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 7
#define TSPMV_DIAG_UNIT_GATHER_SKIP(k) ((k) > 0 && ((d[k].x ^ d[(k) > 0 ? (k) - 1 : 0].x) & 0xFFFFFFu) == 0)
#else
#define TSPMV_DIAG_UNIT_GATHER_SKIP(k) false
#endif

// TILESPMV_ABL 8 (round 6): the classic units' x gathers all fall into the first 256 elements of x (what the launch fetches apart from x: FETCH_SIZE of this build against the product's);
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 8
#define TSPMV_DIAG_UNIT_X(i) ((i) & 255)
#else
#define TSPMV_DIAG_UNIT_X(i) (i)
#endif
// TILESPMV_ABL 9: the short per-strip entry lists (entry mode 0, up to 16 entries with the prologue) gather inside the first 2 KB of x; 10: they are skipped altogether
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 9
#define TSPMV_DIAG_COO0_X(c) ((c) & 255)
#else
#define TSPMV_DIAG_COO0_X(c) (c)
#endif
#if defined(TILESPMV_ABL) && TILESPMV_ABL == 10
#define TSPMV_DIAG_COO0_ON false
#else
#define TSPMV_DIAG_COO0_ON true
#endif

// ---- pooled units (round 5): TILESPMV_POOL_ABL 1 plain LDS store instead of the atomic add, 2 atomic add to a lane-private address (no two lanes of a unit share one), 3 no LDS operation at all
#if defined(TILESPMV_POOL_ABL) && TILESPMV_POOL_ABL == 1
#define TSPMV_DIAG_POOL_ADD(dest, prod) (((&s_y[g][0][0])[dest] = (lacc_t)(prod)), true)
#elif defined(TILESPMV_POOL_ABL) && TILESPMV_POOL_ABL == 2
#define TSPMV_DIAG_POOL_ADD(dest, prod) (atomicAdd(&s_y[g][0][0] + (((dest) & ~15u) | (unsigned)r), (lacc_t)(prod)), true)
#elif defined(TILESPMV_POOL_ABL) && TILESPMV_POOL_ABL == 3
#define TSPMV_DIAG_POOL_ADD(dest, prod) ((acc += (prod)), true)
#else
#define TSPMV_DIAG_POOL_ADD(dest, prod) false
#endif

// TILESPMV_POOL_ABL 4: no x gather (the value is made from the descriptor word: the loads of values, descriptors and row nibbles stay), 6: the gather reads x at the window's first column only (one line per strip)
#if defined(TILESPMV_POOL_ABL) && TILESPMV_POOL_ABL == 4
#define TSPMV_DIAG_POOL_X(load, d) ((val_t)((d).x & 3u))
#elif defined(TILESPMV_POOL_ABL) && TILESPMV_POOL_ABL == 6
#define TSPMV_DIAG_POOL_X(load, d) (x[min((long long)((d).x & POOL_BASE_MASK), xlast)])
#else
#define TSPMV_DIAG_POOL_X(load, d) (load)
#endif

// ---- k_entries_xcd: XCD_ABL 1 no adds to y, 2 no zeroing of the slab, 3 neither and no trip at all (the skeleton: ranges, tasks, barriers)
#if defined(XCD_ABL) && XCD_ABL == 2
#define TSPMV_DIAG_XCD_ZERO 0
#else
#define TSPMV_DIAG_XCD_ZERO 1
#endif
#if defined(XCD_ABL) && XCD_ABL == 3
#define TSPMV_DIAG_XCD_TRIP 0
#else
#define TSPMV_DIAG_XCD_TRIP 1
#endif
#if defined(XCD_ABL) && (XCD_ABL == 1 || XCD_ABL == 3)
#define TSPMV_DIAG_XCD_SKIP_ADDS if (s_acc[tid] == (lacc_t)1.2345e300) y[0] = 1; return;
#else
#define TSPMV_DIAG_XCD_SKIP_ADDS
#endif

// ---- clock stamps (scripts/stamps_probe.py): lane 0 of every wavefront records the shader clock at a few points of k_units; the stamps go to a buffer of their own and no
// output depends on them.  Read the SHARES, not the length (the waits the stamps force are not in the real kernel).
#ifdef TILESPMV_STAMPS
#define TSPMV_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); stamp_[7] = rt_; }
#define TSPMV_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0); stamp_[i] = t_; } while (0)
#define TSPMV_STAMP_WAIT(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TSPMV_STAMP(i); } while (0)
#define TSPMV_STAMP_STORE TSPMV_STAMP(5); TSPMV_STAMP_WAIT(6); \
    if (S.stamps && (tid & 63) == 0) { unsigned long long *o = S.stamps + ((long long)blockIdx.x * 4 + (tid >> 6)) * 8; for (int i = 0; i < 8; i++) o[i] = stamp_[i]; }
#else
#define TSPMV_STAMP_DECL
#define TSPMV_STAMP(i) do { } while (0)
#define TSPMV_STAMP_WAIT(i) do { } while (0)
#define TSPMV_STAMP_STORE
#endif
