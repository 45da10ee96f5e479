// hip_plan_internal.h — what the plan's translation units share: the resolved knobs, the plan object with its device arena, and the
// per-tile repacking helpers.  hip_plan.hip holds the C ABI, the first-generation layout, the fallback lists and the measured
// selection; hip_plan_stream.hip holds the second-generation ("unit stream") layout builder, in stages.
#pragma once
#include <hip/hip_runtime.h>
#include <sys/time.h>

#include <climits>
#include <cmath>
#include <string>
#include <unordered_map>
#include <unordered_set>

#include "hip_plan.h"
#include "plan_tile_ops.h"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "tilespmv: HIP error %d (%s) at %s:%d: %s\n", (int)e_, hipGetErrorString(e_), \
                    __FILE__, __LINE__, #expr);                                                         \
            return (int)e_;                                                                             \
        }                                                                                               \
    } while (0)

namespace tilespmv {

inline double now_us()
{
    timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec * 1e6 + t.tv_usec;
}

inline int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// Every tuning knob of one plan build, resolved ONCE at the API boundary (tilespmv_plan_create): option field if set, else the
// environment variable (getenv only — the library never writes the environment), else the built-in default.  The builder and
// the autotuner pass this struct around; nothing below the boundary reads the environment.
struct Knobs {
    int coo_mode, dense_mode, kernel, tilerow_begin, tilerow_end, autotune;
    int entry_mode;      // -1 = chosen from the shard
    int entry_ordered;   // -1 = chosen from the grid size
    int strip_cost;      // <= 0 = chosen from the shard
    int split_above, split_cap, xcd_remap, xcd_chunk, csr_split, fix_inline, coo_cost, coo_heavy_min;
    int coo_piece;       // <= 0 = derived from the piece size
    int strip_even;
    int wg_strips;       // -1 = chosen from the shard
    int x_window;        // -1 = default
    int x_stride1, x_stride2;   // tile-rows per grid line / plane for the x windows; 0 = detected from the shard
    int lds_pad;         // bytes of unused LDS added to every unit-kernel workgroup (fewer resident workgroups per CU); -1 = chosen from the shard
    int brick_rows;      // brick order: tile-rows per strip at most (experiment knob, environment only)
    int y_store;         // -1 by rule, 0 plain y stores, 1 streaming (nontemporal) y stores
    int mv_native;       // -1 = by nvec
    int mv_xcd_chunk;    // -1 = the plan's XCD chunk
    int nt_stream;       // -1 by size, 0 plain, 1 nontemporal loads of the value / entry-record streams
    int x_panel_kb;      // column panels of the entry lists: KB of x per panel; 0 off, -1 by rule
    int x_panel_merge;   // ... panels per pass; 0 unpanelled launch, -1 chosen by timing
    int x_slice_passes;  // column slices pinned to XCDs: passes (8 slices each); 0 off, -1 chosen by timing beside the panelled forms
    int placement_tries; // large plans: arena placements timed at plan creation (-1 by size, 0 / 1 off, n)
    int deterministic;   // 1: nothing is chosen by timing, every sum in a plan-fixed order (tilespmv_plan_options.deterministic)
    int absorb;          // 1 (default): list entries next to an ELL tile go into its padding slots (plan_tile_ops.h "absorbed list entries"); 0 = never
    int desc_dict;       // 0 = always 12-B unit descriptors; -1 = 4-B descriptors + pattern dictionary where the shard allows and it pays; 1 = wherever it allows; 2 = like 1 (pooled plans: 8-byte pairs instead of 4-byte words)
    bool xcd_from_caller, entry_from_caller, strip_from_caller;   // the autotuner leaves alone what the caller pinned
    bool dry;            // tilespmv_plan_layout_digest: build the layout on the host only, hash instead of upload
    const char *autotune_log;
};

}  // namespace tilespmv

using namespace tilespmv;

// k_entries_xcd's trip size for `passes` slice passes: the smallest of 4 / 6 / 8 records per lane whose trip (x 256 lanes) holds an average run with 10 % to spare
static inline int slice_trip_records(long long list_records, int groups, int passes)
{
    const double avg = (double)list_records / std::max(1.0, 8.0 * passes * (double)std::max(1, groups));
    return avg * 1.1 <= 1024 ? 4 : avg * 1.1 <= 1536 ? 6 : 8;
}

struct tilespmv_plan {
    DevPlan dev{};
    DevStream st{};
    DevDense dn{};
    int xcd_remap = 2, xcd_chunk = 32;  // windows of 8 x 32 workgroups: neighbouring strips share an XCD L2 (sweep 4..64: flat within 2.5 %, 32 best on 3 of 4 large matrices)
    val_t *mv_x = nullptr, *mv_y = nullptr;  // plans without a native multi-vector kernel: X / Y as mv_nvec contiguous vectors (allocated at the first such call)
    int mv_nvec = 0;
    bool mv_by_columns = false;              // ... and plans whose work is mostly COO entries (the multi-vector kernel walks them per 16-lane strip)
    int mv_native = -1, mv_xcd_chunk = -1;   // knobs of tilespmv_plan_spmm (Knobs)
    int mv_slab_rows = 0;                    // > 0: the multi-vector kernel scatters a strip's entries up front into an LDS slab of this many tile-rows per lane group
    bool pooled = false;                // every unit is a pooled unit (hip_plan.h): k_units<.., POOL>, k_pool_mv
    int entry_mode = 0;                 // COO entry lists walked per 16-lane strip (0), per wavefront (1) or per workgroup, column-ordered (2)
    std::vector<void *> allocs;
    std::vector<std::pair<void *, size_t>> arena_blocks;   // the blocks the plan's streams were carved from (and the partial-slot array), with their sizes: what a re-placement moves
    long long info[TILESPMV_INFO_COUNT] = {0};
    int coo_mode = 0, dense_mode = 0, kernel = 0;
    int device = 0;
    int wg_strips = 16;                 // strips per workgroup of the unit kernel (32 only with the workgroup entry mode)
    int lds_pad_bytes = 0;              // extra (unused) dynamic LDS per workgroup of the unit kernel: caps the workgroups resident on a CU (knob lds_pad)
    int arena_flags = 0; size_t arena_skew = 0, arena_spacer = 0; bool arena_spacer_first_only = false;
    // (Round 4's opt-in chunked physical backing of large blocks — hipMemAddressReserve / hipMemCreate / hipMemMap, TILESPMV_ARENA_VMM_MB — changed the placement state for the better in two of
    // four sessions and had to keep its virtual ranges reserved for the life of the process because of stale translations on re-reserved ranges (LABBOOK S6.19, scripts/micro/vmm_probe.hip): retired in round 6.)
    // a block of the arena (or a candidate placement of one); the plan owns it until block_free / destroy
    int block_alloc(void **out, size_t bytes, bool quiet = false)
    {
        // A block starts out as zeros, explicitly: the kernels read a little past the end of some streams (a strip's last descriptor chunk, masked tail lanes — the 256 bytes of
        // slack behind every stream are there for that) and what they find must decode to "nothing" (unit 0, offset 0).  hipMalloc happens to hand out zeroed memory;
        // other allocators do not (found the hard way: stale descriptors behind the last strip sent a value prefetch to a wild address).
        *out = nullptr;
        hipError_t e = arena_flags ? hipExtMallocWithFlags(out, bytes, (unsigned)arena_flags) : hipMalloc(out, bytes);   // experiment knob TILESPMV_ARENA_FLAGS (4 = physically contiguous)
        if (e == hipSuccess) e = hipMemset(*out, 0, bytes);
        if (e != hipSuccess) {
            // a tolerated failure (a candidate placement that does not fit) must not leave HIP's sticky last error behind: the next launch_* returns hipGetLastError() and
            // would report this out-of-memory for a healthy launch
            (void)hipGetLastError();
            if (!quiet) fprintf(stderr, "tilespmv: device allocation of %zu MB failed: %s\n", bytes >> 20, hipGetErrorString(e));
            if (*out) (void)hipFree(*out);
            *out = nullptr;
            return -1;
        }
        allocs.push_back(*out);
        return 0;
    }
    void block_free(void *p)
    {
        (void)hipFree(p);
        auto it = std::find(allocs.begin(), allocs.end(), p);
        if (it != allocs.end()) allocs.erase(it);
    }
    size_t arena_used = 0;              // bytes handed out by upload() so far
    char *arena_at = nullptr; size_t arena_left = 0, arena_block = (size_t)256 << 20, arena_next = (size_t)1 << 20, size_hint = 0;   // bump allocator of upload(); size_hint = the builder's estimate of the plan's bytes
    bool dry = false;                   // layout-digest build: no HIP call, streams are hashed instead of uploaded
    long long list_records = 0;         // records of the merged entry lists (workgroup entry mode)
    bool slice_calibrate = false;       // ... and whether column slices pinned to XCDs beat them (DevStream::slice_passes)
    bool panel_calibrate = false;       // panels recorded, panels per pass still to be chosen by timing (plan_create_one)
    long long panel_rmw_rows = 0;       // rows of y the passes beyond the first read and write, at the finest panels (byte model)
    unsigned long long digest = 1469598103934665603ull;
    unsigned long long stage_digest[TILESPMV_STAGE_COUNT] = {0};   // layout-digest builds: one hash per stage of the unit-stream builder (hip_plan_stream.hip)
    std::vector<size_t> uploaded_bytes;          // ... and the bytes of each (tilespmv_plan_stream_digests)
    std::vector<const void **> uploaded_slots;   // every device-pointer MEMBER of this plan that upload() filled: what a re-placement rebases (hip_plan.hip retry_placement)
    // arena space for n elements that a device kernel will fill (zeroed like every block); same bookkeeping as upload()
    template <class T>
    int reserve(size_t n, const T **out) { return upload<T>(nullptr, n, out); }
    template <class T>
    int upload(const T *host, size_t n, const T **out)
    {
        if ((const char *)out >= (const char *)this && (const char *)out < (const char *)this + sizeof(*this)) { uploaded_slots.push_back((const void **)out); uploaded_bytes.push_back(n * sizeof(T)); }
        if (dry) {   // FNV-1a-64 over (element count, bytes) of every stream, in upload order
            auto mix = [&](const unsigned char *p, size_t len) { for (size_t i = 0; i < len; i++) { digest ^= p[i]; digest *= 1099511628211ull; } };
            const unsigned long long cnt = n;
            mix((const unsigned char *)&cnt, 8);
            mix((const unsigned char *)host, n * sizeof(T));
            info[TILESPMV_INFO_DEVICE_BYTES] += (long long)(n * sizeof(T));
            *out = nullptr;
            return 0;
        }
        const double t0 = now_us();
        void *d = nullptr;
        // Streams are carved out of a few large device blocks (bump allocation, 256-byte aligned + 256 bytes of slack so that
        // masked tail lanes never fault) instead of one hipMalloc each: a plan is ~20 streams, and large blocks get large
        // page-table fragments whatever state the allocator is in (fewer hipMalloc calls, too).
        const size_t need = (std::max<size_t>(n, 1) * sizeof(T) + 256 + 255) / 256 * 256 + arena_skew;   // (arena_skew: experiment knob, bytes left unused behind every stream)
        if (need > arena_left) {   // blocks of arena_block bytes (256 MB) for plans of that size and more; a smaller plan gets one block of about its own size (size_hint)
            // ... and the LAST blocks of a large plan are sized by what the builder still expects to upload (round 5: config 4's plan is 689 MB — its third 256-MB block was
            // followed by a fourth for the last few MB of task records, and allocating + zeroing that block was 90 ms of a 260-ms plan creation)
            const size_t left_hint = size_hint > arena_used ? size_hint - arena_used : 0;
            const size_t tail = std::max<size_t>((size_t)32 << 20, left_hint + left_hint / 8 + ((size_t)4 << 20));
            const size_t want = size_hint >= arena_block ? std::min(arena_block, tail) : std::max<size_t>(arena_next, size_hint + size_hint / 4 + ((size_t)1 << 20));
            const size_t blk = std::max<size_t>(need, std::min(want, arena_block));
            arena_next = std::min<size_t>(arena_next * 4, std::max<size_t>(arena_block, 1));
            void *b = nullptr;
            if (arena_spacer && (!arena_spacer_first_only || arena_blocks.empty())) {   // experiment knob TILESPMV_ARENA_SPACER_MB: an unused allocation in front of every block (does where a block lands decide its state? DESIGN.md S6.19)
                void *sp = nullptr;
                if (hipMalloc(&sp, arena_spacer) == hipSuccess) allocs.push_back(sp); else (void)hipGetLastError();
            }
            if (block_alloc(&b, blk) != 0) return -1;
            arena_blocks.push_back({b, blk});
            arena_at = (char *)b; arena_left = blk;
        }
        d = arena_at; arena_at += need; arena_left -= need; arena_used += need;
        if (n && host) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
        info[TILESPMV_INFO_DEVICE_BYTES] += (long long)(n * sizeof(T));
        info[TILESPMV_INFO_UPLOAD_US] += (long long)(now_us() - t0);
        *out = (const T *)d;
        return 0;
    }
};

namespace tilespmv {

inline int nib(const unsigned char *s, long long p) { return (p & 1) ? (s[p >> 1] & 15) : (s[p >> 1] >> 4); }
inline void put_nib(unsigned char *s, int p, int v) { if (p & 1) s[p >> 1] |= (unsigned char)v; else s[p >> 1] |= (unsigned char)(v << 4); }

// What one source tile becomes in the streams.
struct Emit { int fmt, p1, p2, nv, ni; };

inline Emit emit_of(const Tile_matrix *T, int t, int rowlen, bool coo_in_tile)
{
    Emit e{DESC_FMT_NOP, 0, 0, 0, 0};
    const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
    switch (fmt) {
    case TILESPMV_FMT_CSR: e.fmt = fmt; e.p1 = stored; break;
    case TILESPMV_FMT_COO: if (!coo_in_tile) return e; e.fmt = fmt; e.p1 = stored; break;
    case TILESPMV_FMT_ELL: e.fmt = fmt; e.p1 = w; break;
    case TILESPMV_FMT_HYB: e.fmt = fmt; e.p1 = w; e.p2 = coo_in_tile ? stored - w * rowlen : 0; break;
    case TILESPMV_FMT_DNS: e.fmt = fmt; break;
    case TILESPMV_FMT_DNSROW: e.fmt = fmt; e.p1 = T->dnsrowptr[t + 1] - T->dnsrowptr[t]; break;
    case TILESPMV_FMT_DNSCOL: e.fmt = fmt; e.p1 = T->dnscolptr[t + 1] - T->dnscolptr[t]; break;
    }
    tile_stream_sizes(e.fmt, e.p1, e.p2, &e.nv, &e.ni);
    return e;
}

// Copy one tile's payload into the streams, converting to row stride 16 / tile-local packing.
inline void repack_tile(const Tile_matrix *T, int t, const Emit &e, int rowlen, int collen, long long hyb_idx_off,
                 val_t *v, unsigned char *ix)
{
    switch (e.fmt) {
    case TILESPMV_FMT_CSR: {
        const int off = T->csr_offset[t], poff = T->csrptr_offset[t];
        memcpy(v, T->Blockcsr_Val + off, sizeof(val_t) * (size_t)e.p1);
        for (int r = 0; r < 16; r++) ix[r] = (unsigned char)(r < rowlen ? T->Blockcsr_Ptr[poff + r] : e.p1);
        for (int k = 0; k < e.p1; k++) put_nib(ix + 16, k, nib(T->csr_compressedIdx, (long long)off + k));
        break;
    }
    case TILESPMV_FMT_COO: {
        const int off = T->coo_offset[t];
        memcpy(v, T->Blockcoo_Val + off, sizeof(val_t) * (size_t)e.p1);
        memcpy(ix, T->coo_compressed_Idx + off, (size_t)e.p1);
        break;
    }
    case TILESPMV_FMT_ELL: {
        const int off = T->ell_offset[t];
        for (int s = 0; s < e.p1; s++)
            for (int r = 0; r < rowlen; r++) {
                v[16 * s + r] = T->Blockell_Val[off + s * rowlen + r];
                put_nib(ix, 16 * s + r, nib(T->ell_compressedIdx, (long long)off + s * rowlen + r));
            }
        break;
    }
    case TILESPMV_FMT_HYB: {
        const int off = T->hyb_offset[t], nell = e.p1 * rowlen;
        const unsigned char *src = T->hybIdx + hyb_idx_off;
        for (int s = 0; s < e.p1; s++)
            for (int r = 0; r < rowlen; r++) {
                v[16 * s + r] = T->Blockhyb_Val[off + s * rowlen + r];
                put_nib(ix, 16 * s + r, nib(src, s * rowlen + r));
            }
        for (int i = 0; i < e.p2; i++) {
            v[16 * e.p1 + i] = T->Blockhyb_Val[off + nell + i];
            ix[8 * e.p1 + i] = src[(nell + 1) / 2 + i];
        }
        break;
    }
    case TILESPMV_FMT_DNS: {
        const int off = T->dns_offset[t];
        for (int c = 0; c < collen; c++)
            for (int r = 0; r < rowlen; r++) v[16 * c + r] = T->Blockdense_Val[off + c * rowlen + r];
        break;
    }
    case TILESPMV_FMT_DNSROW: {
        const int off = T->dnsrow_offset[t], ro = T->dnsrowptr[t];
        for (int k = 0; k < e.p1; k++) {
            for (int c = 0; c < collen; c++) v[16 * k + c] = T->Blockdenserow_Val[off + k * collen + c];
            ix[k] = (unsigned char)T->denserowid[ro + k];
        }
        break;
    }
    case TILESPMV_FMT_DNSCOL: {
        const int off = T->dnscol_offset[t], co = T->dnscolptr[t];
        for (int k = 0; k < e.p1; k++) {
            for (int r = 0; r < rowlen; r++) v[16 * k + r] = T->Blockdensecol_Val[off + k * rowlen + r];
            ix[k] = (unsigned char)T->densecolid[co + k];
        }
        break;
    }
    default: break;
    }
}


// One entry of a merged list before packing.
struct PEnt { unsigned col, dest; val_t val; };

// Packs one list (entries already in their final order: by column, ties in list order) into records and per-chunk column
// bases (hip_plan.h ERec).  Chunk k of the list = its records [64k, 64k + 64); base = column of the chunk's first entry;
// an entry whose column is 2^(32 - dest_bits) or more above the base closes the chunk, which is filled up with null
// records (value 0, offset 0, destination 0: adds 0 * x[base] to the group's first row).  Returns false if the packed list
// does not decode back to the input (checked in layout-digest builds).
inline bool pack_list(const std::vector<PEnt> &ents, int dest_bits, std::vector<ERec> &rec, std::vector<unsigned> &base, bool verify)
{
    const size_t rec0 = rec.size(), base0 = base.size();
    pack_chunks((long long)ents.size(), dest_bits, [&](long long i) { return ents[(size_t)i].col; },
                [&](long long i, unsigned b) { rec.push_back(make_erec(ents[(size_t)i].val, ((ents[(size_t)i].col - b) << dest_bits) | ents[(size_t)i].dest)); },
                [&]() { rec.push_back(make_erec((val_t)0, 0u)); }, [&](unsigned b) { base.push_back(b); });   // (plan_tile_ops.h: shared with the device builder)
    if (!verify) return true;
    size_t j = 0;
    for (size_t q = rec0; q < rec.size(); q++) {
        const ERec &r = rec[q];
        const unsigned bq = base[base0 + (q - rec0) / ECHUNK];
        val_t v;
#if defined(TILESPMV_F32)
        memcpy(&v, &r.v, 4);
#else
        unsigned bb[2] = {r.lo, r.hi}; memcpy(&v, bb, 8);
#endif
        if (r.w == 0u && v == (val_t)0 && (j >= ents.size() || ents[j].col != bq || ents[j].dest != 0u || ents[j].val != (val_t)0)) continue;   // null padding
        if (j >= ents.size()) return false;
        const unsigned col = bq + (r.w >> dest_bits), dest = r.w & ((1u << dest_bits) - 1u);
        if (col != ents[j].col || dest != ents[j].dest || memcmp(&v, &ents[j].val, sizeof(val_t)) != 0) return false;
        j++;
    }
    return j == ents.size() && base.size() - base0 == (rec.size() - rec0 + ECHUNK - 1) / ECHUNK;
}

// the value pass of the ENCODE stage on the device (hip_kernels.hip k_pair_values)
hipError_t launch_pair_values(const val_t *src, val_t *dst, const int4 *map, int ntasks);


// Second-generation layout (hip_plan_stream.hip): fills plan->st / plan->dn / the whole-tile pass of plan->dev for tile-rows [tr0, tr1).
int build_stream(tilespmv_plan *plan, const Knobs &K, const Tile_matrix *T, int rowA, int colA, int tr0, int tr1, bool coo_in_tile,
                 bool dense_mfma, const std::vector<long long> &hyb_off,
                 std::vector<FixRow> &fix, int &npartial, long long &n_tasks, long long &model_bytes, const struct DevTile *DT = nullptr);   // DT: device mode (hip_plan_device.h)

}  // namespace tilespmv
