// hip_plan.hip — builds and owns the device-resident plan; C ABI of the GPU hot path.
//
// Replaces the host half of call_tilespmv_cuda (reference src/tilespmv_cuda.h:794-1180): instead
// of ~35 cudaMalloc+cudaMemcpy pairs of the per-format arrays (:867-1004) the Tile_matrix is
// re-laid-out once into tile-ordered streams (hip_plan.h: the unit stream of generation 2, or the
// tile stream of generation 1) and uploaded; the chunk schedule that the reference derives inside
// tilespmv_cpu (:68-118) and patches up with a one-off v5 launch (:1045-1056) is replaced by a
// cost-balanced strip list built here.
#include <hip/hip_runtime.h>
#include <sys/time.h>

#include <climits>
#include <cmath>
#include <string>
#include <unordered_map>
#include <unordered_set>

#include "hip_plan.h"

namespace tilespmv {

hipError_t launch_tiles_direct(const DevPlan &P, bool dense_mfma, bool accumulate, bool fixup, const val_t *x, val_t *y, hipStream_t st);
hipError_t launch_tiles_stream(const DevPlan &P, const DevStream &S, const DevDense &DN, bool dense_mfma, int entry_mode, int wg_strips, int xwin_lds_bytes, int lds_pad_bytes, int xcd_remap, int xcd_chunk,
                               const val_t *x, val_t *y, hipStream_t st);
hipError_t launch_fallback(const DevPlan &P, const val_t *x, val_t *y, hipStream_t st);
hipError_t launch_tiles_stream_mv(const DevPlan &P, const DevStream &S, const DevDense &DN, int nvec, int xcd_chunk, bool entries_pass, int slab_rows, const val_t *X, val_t *Y,
                                  hipStream_t st);
hipError_t launch_rows_to_columns(const val_t *X, int nvec, long long n, long long ld, val_t *XT, hipStream_t st);
hipError_t launch_columns_to_rows(const val_t *YT, int nvec, long long row0, long long rows, long long ld, val_t *Y, hipStream_t st);

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "tilespmv: HIP error %d (%s) at %s:%d: %s\n", (int)e_, hipGetErrorString(e_), \
                    __FILE__, __LINE__, #expr);                                                         \
            return (int)e_;                                                                             \
        }                                                                                               \
    } while (0)

static double now_us()
{
    timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec * 1e6 + t.tv_usec;
}

static int env_int(const char *name, int dflt)
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
    int desc_dict;       // 0 = always 12-B unit descriptors; -1 = 4-B descriptors + pattern dictionary where the shard allows and it pays; 1 = wherever it allows
    bool xcd_from_caller, entry_from_caller, strip_from_caller;   // the autotuner leaves alone what the caller pinned
    bool dry;            // tilespmv_plan_layout_digest: build the layout on the host only, hash instead of upload
    const char *autotune_log;
};

static Knobs resolve_knobs(const tilespmv_plan_options *opts)
{
    tilespmv_plan_options o;
    tilespmv_plan_options_init(&o);
    if (opts && opts->size >= sizeof(unsigned)) memcpy(&o, opts, std::min<size_t>(opts->size, sizeof(o)));
    o.size = (unsigned)sizeof(o);
    auto pick = [](int opt, const char *env, int dflt) { return opt != TILESPMV_KNOB_DEFAULT ? opt : env_int(env, dflt); };
    auto pinned = [](int opt, const char *env) { return opt != TILESPMV_KNOB_DEFAULT || getenv(env) != nullptr; };
    Knobs k{};
    k.coo_mode = o.coo_mode > 0 ? o.coo_mode : env_int("TILESPMV_COO_MODE", 0);
    k.dense_mode = o.dense_mode > 0 ? o.dense_mode : env_int("TILESPMV_DENSE_MODE", 0);
    k.kernel = o.kernel > 0 ? o.kernel : env_int("TILESPMV_KERNEL", 0);
    k.tilerow_begin = o.tilerow_begin; k.tilerow_end = o.tilerow_end;
    k.autotune = (o.autotune > 0 || env_int("TILESPMV_AUTOTUNE", 0) != 0) ? 1 : 0;
    k.entry_mode = pick(o.entry_mode, "TILESPMV_WAVE_COO", -1);
    k.entry_ordered = pick(o.entry_ordered, "TILESPMV_COO_ORDERED", -1);
    k.strip_cost = pick(o.strip_cost, "TILESPMV_STRIP_COST", 0);
    k.split_above = pick(o.split_above, "TILESPMV_SPLIT_ABOVE", 2400);
    k.split_cap = pick(o.split_cap, "TILESPMV_SPLIT_CAP", 4800);
    k.xcd_remap = pick(o.xcd_remap, "TILESPMV_XCD_REMAP", 2) ? 2 : 0;
    k.xcd_chunk = std::max(1, pick(o.xcd_chunk, "TILESPMV_XCD_CHUNK", 32));
    k.csr_split = pick(o.csr_split, "TILESPMV_CSR_SPLIT", 1);
    k.fix_inline = pick(o.fix_inline, "TILESPMV_FIX_INLINE", 1);
    k.coo_cost = pick(o.coo_cost, "TILESPMV_COO_COST", 4);
    k.coo_heavy_min = std::max(0, pick(o.coo_heavy_min, "TILESPMV_COO_HEAVY_MIN", 32));
    k.coo_piece = pick(o.coo_piece, "TILESPMV_COO_PIECE", 0);
    k.strip_even = pick(o.strip_even, "TILESPMV_STRIP_EVEN", 4);
    k.wg_strips = pick(o.wg_strips, "TILESPMV_WG_STRIPS", -1);
    k.x_window = pick(o.x_window, "TILESPMV_X_WINDOW", -1);
    k.x_stride1 = pick(o.x_stride1, "TILESPMV_X_STRIDE1", 0);
    k.x_stride2 = pick(o.x_stride2, "TILESPMV_X_STRIDE2", 0);
    k.lds_pad = pick(o.lds_pad, "TILESPMV_LDS_PAD", -1);
    k.brick_rows = env_int("TILESPMV_BRICK_ROWS", 0);
    k.y_store = pick(o.y_store, "TILESPMV_Y_STORE", -1);
    k.mv_native = pick(o.mv_native, "TILESPMV_MV_NATIVE", -1);
    k.mv_xcd_chunk = pick(o.mv_xcd_chunk, "TILESPMV_MV_XCD_CHUNK", -1);
    k.desc_dict = pick(o.desc_dict, "TILESPMV_DESC_DICT", -1);
    k.nt_stream = pick(o.nt_stream, "TILESPMV_NT_STREAM", -1);
    k.xcd_from_caller = pinned(o.xcd_remap, "TILESPMV_XCD_REMAP") || pinned(o.xcd_chunk, "TILESPMV_XCD_CHUNK");
    k.entry_from_caller = pinned(o.entry_mode, "TILESPMV_WAVE_COO");
    k.strip_from_caller = o.strip_cost > 0 || env_int("TILESPMV_STRIP_COST", 0) > 0;
    k.autotune_log = getenv("TILESPMV_AUTOTUNE_LOG");
    return k;
}

}  // namespace tilespmv

using namespace tilespmv;

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
    int entry_mode = 0;                 // COO entry lists walked per 16-lane strip (0), per wavefront (1) or per workgroup, column-ordered (2)
    std::vector<void *> allocs;
    long long info[TILESPMV_INFO_COUNT] = {0};
    int coo_mode = 0, dense_mode = 0, kernel = 0;
    int device = 0;
    int wg_strips = 16;                 // strips per workgroup of the unit kernel (32 only with the workgroup entry mode)
    int lds_pad_bytes = 0;              // extra (unused) dynamic LDS per workgroup of the unit kernel: caps the workgroups resident on a CU (knob lds_pad)
    int xwin_lds_bytes = 0;             // x-window plans: dynamic LDS of the unit kernel (largest window of the plan); 0 = no windows
    int arena_flags = 0; size_t arena_skew = 0;
    char *arena_at = nullptr; size_t arena_left = 0, arena_block = (size_t)256 << 20, arena_next = (size_t)1 << 20, size_hint = 0;   // bump allocator of upload(); size_hint = the builder's estimate of the plan's bytes
    bool dry = false;                   // layout-digest build: no HIP call, streams are hashed instead of uploaded
    unsigned long long digest = 1469598103934665603ull;
    template <class T>
    int upload(const T *host, size_t n, const T **out)
    {
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
            const size_t want = size_hint >= arena_block ? arena_block : std::max<size_t>(arena_next, size_hint + size_hint / 4 + ((size_t)1 << 20));
            const size_t blk = std::max<size_t>(need, std::min(want, arena_block));
            arena_next = std::min<size_t>(arena_next * 4, std::max<size_t>(arena_block, 1));
            void *b = nullptr;
            if (arena_flags) HIP_TRY(hipExtMallocWithFlags(&b, blk, (unsigned)arena_flags));   // experiment knob TILESPMV_ARENA_FLAGS (4 = physically contiguous)
            else HIP_TRY(hipMalloc(&b, blk));
            allocs.push_back(b);
            arena_at = (char *)b; arena_left = blk;
        }
        d = arena_at; arena_at += need; arena_left -= need;
        if (n) HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
        info[TILESPMV_INFO_DEVICE_BYTES] += (long long)(n * sizeof(T));
        info[TILESPMV_INFO_UPLOAD_US] += (long long)(now_us() - t0);
        *out = (const T *)d;
        return 0;
    }
};

namespace {

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
void repack_tile(const Tile_matrix *T, int t, const Emit &e, int rowlen, int collen, long long hyb_idx_off,
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

}  // namespace


// ------------------------------------------------------------------------------------------------
// Second-generation layout builder (hip_plan.h "unit stream").
// ------------------------------------------------------------------------------------------------
namespace {

// One entry of a merged list before packing.
struct PEnt { unsigned col, dest; val_t val; };

inline ERec make_erec(val_t v, unsigned w)
{
    ERec r;
#if defined(TILESPMV_F32)
    memcpy(&r.v, &v, 4);
#else
    unsigned b[2]; memcpy(b, &v, 8); r.lo = b[0]; r.hi = b[1];
#endif
    r.w = w;
    return r;
}

// Packs one list (entries already in their final order: by column, ties in list order) into records and per-chunk column
// bases (hip_plan.h ERec).  Chunk k of the list = its records [64k, 64k + 64); base = column of the chunk's first entry;
// an entry whose column is 2^(32 - dest_bits) or more above the base closes the chunk, which is filled up with null
// records (value 0, offset 0, destination 0: adds 0 * x[base] to the group's first row).  Returns false if the packed list
// does not decode back to the input (checked in layout-digest builds).
inline bool pack_list(const std::vector<PEnt> &ents, int dest_bits, std::vector<ERec> &rec, std::vector<unsigned> &base, bool verify)
{
    const unsigned long long span = 1ull << (32 - dest_bits);
    const size_t rec0 = rec.size(), base0 = base.size();
    size_t i = 0;
    while (i < ents.size()) {
        const unsigned b = ents[i].col;
        base.push_back(b);
        int n = 0;
        while (i < ents.size() && n < ECHUNK && (unsigned long long)ents[i].col - b < span) {
            rec.push_back(make_erec(ents[i].val, ((ents[i].col - b) << dest_bits) | ents[i].dest));
            i++; n++;
        }
        if (i < ents.size()) for (; n < ECHUNK; n++) rec.push_back(make_erec((val_t)0, 0u));   // interior chunks are always full
    }
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

struct RowCount { int nunits, ncoo, nheavy, ndense; long long hval, hidx; long long cost; };

// A CSR tile is executed as w ELL-style units (the first w entries of every row) plus the rest of
// its entries on the strip's COO list; w minimises the bytes moved (HYB's idea, src/csr2tile.h:279-306,
// with this kernel's byte costs).  Returns w and the number of remainder entries.
inline int csr_split_width(const unsigned char *ptr, int rowlen, int nnz, int *remainder)
{
    const long long unit_b = 16 + 16 * (long long)sizeof(val_t), entry_b = (long long)sizeof(val_t) + 5;
    int len[16], wmax = 0;
    for (int r = 0; r < 16; r++) { len[r] = r < rowlen ? ((r == rowlen - 1 ? nnz : ptr[r + 1]) - ptr[r]) : 0; wmax = std::max(wmax, len[r]); }
    int best_w = 0, best_rem = nnz; long long best = entry_b * nnz;
    for (int w = 1; w <= wmax; w++) {
        int rem = 0;
        for (int r = 0; r < 16; r++) rem += std::max(0, len[r] - w);
        const long long b = unit_b * w + entry_b * rem;
        if (b < best) { best = b; best_w = w; best_rem = rem; }
    }
    *remainder = best_rem;
    return best_w;
}

inline RowCount count_row(const Tile_matrix *T, int bi, int rowlen, int tilen, int colA, bool coo_in_tile, bool dense_mfma, bool csr_split, int coo_cost)
{
    RowCount c{0, 0, 0, 0, 0, 0, 0};
    for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
        const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
        switch (fmt) {
        case TILESPMV_FMT_ELL: c.nunits += w; break;
        case TILESPMV_FMT_HYB: c.nunits += w; if (coo_in_tile) c.ncoo += stored - w * rowlen; break;
        case TILESPMV_FMT_DNSCOL: c.nunits += T->dnscolptr[t + 1] - T->dnscolptr[t]; break;
        case TILESPMV_FMT_DNS:
            if (dense_mfma) c.ndense++;
            else c.nunits += tile_collen(T->tile_columnidx[t], tilen, colA);
            break;
        case TILESPMV_FMT_COO: if (coo_in_tile) c.ncoo += stored; break;
        case TILESPMV_FMT_CSR:
            if (csr_split) { int rem; c.nunits += csr_split_width(T->Blockcsr_Ptr + T->csrptr_offset[t], rowlen, stored, &rem); c.ncoo += rem; }
            else { c.nheavy++; c.hval += stored; c.hidx += 16 + (stored + 1) / 2; }
            break;
        case TILESPMV_FMT_DNSROW: c.nunits += T->dnsrowptr[t + 1] - T->dnsrowptr[t]; break;  // one row unit per dense row
        }
    }
    // heavy tiles are latency-bound (one tile at a time): weigh them so that long lists get split
    c.cost = 16LL * c.nunits + (long long)coo_cost * c.ncoo + c.hval + 256LL * c.nheavy + 64LL * c.ndense + 8;
    return c;
}


// Dominant tile-row distances of a stencil-like shard: d = column block - tile-row over the tiles that become units.  s1 = the
// smallest distance >= 2 that most tile-rows have (tile-rows per grid line), s2 = the middle of the next cluster of distances
// (tile-rows per grid plane; 0 for 2-D problems).  0 / 0 when the shard has no such structure.
inline void detect_strides(const Tile_matrix *T, int tr0, int tr1, bool csr_split, bool dense_mfma, int *s1, int *s2)
{
    *s1 = *s2 = 0;
    const int ntr = tr1 - tr0;
    if (ntr < 32) return;
    const int step = std::max(1, ntr / 32768);
    std::vector<long long> ds;
    long long sampled = 0;
    for (int bi = tr0; bi < tr1; bi += step) {
        sampled++;
        long long last = LLONG_MIN;
        for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
            const int fmt = T->Format[t];
            const bool units = fmt == TILESPMV_FMT_ELL || fmt == TILESPMV_FMT_HYB || fmt == TILESPMV_FMT_DNSCOL || fmt == TILESPMV_FMT_DNSROW ||
                               (fmt == TILESPMV_FMT_DNS && !dense_mfma) || (fmt == TILESPMV_FMT_CSR && csr_split);
            const long long d = (long long)T->tile_columnidx[t] - bi;
            if (units && d >= 2 && d != last) { ds.push_back(d); last = d; }
        }
    }
    std::sort(ds.begin(), ds.end());
    std::vector<long long> dom;   // distances that at least a quarter of the sampled tile-rows have
    for (size_t i = 0; i < ds.size();) {
        size_t j = i;
        while (j < ds.size() && ds[j] == ds[i]) j++;
        if ((long long)(j - i) * 4 >= sampled) dom.push_back(ds[i]);
        i = j;
    }
    if (dom.empty() || dom[0] > (1 << 20)) return;
    *s1 = (int)dom[0];
    size_t a = 1;
    while (a < dom.size() && dom[a] <= dom[0] + 1) a++;
    if (a >= dom.size()) return;
    size_t b = a;
    while (b + 1 < dom.size() && dom[b + 1] - dom[b] <= dom[0] + 1) b++;   // {s2 - s1, s2, s2 + s1} of a 27-point stencil
    const long long mid = dom[(a + b) / 2];
    if (mid % dom[0] == 0 && mid / dom[0] >= 2 && mid < (1ll << 30)) *s2 = (int)mid;
}

}  // namespace

static int build_stream(tilespmv_plan *plan, const Knobs &K, const Tile_matrix *T, int rowA, int colA, int tr0, int tr1, bool coo_in_tile,
                        bool dense_mfma, const std::vector<long long> &hyb_off,
                        std::vector<FixRow> &fix, int &npartial, long long &n_tasks, long long &model_bytes)
{
    const bool csr_split = K.csr_split != 0;
    const int target_in = K.strip_cost, split_above_in = K.split_above;
    const int tilem = T->tilem, tilen = T->tilen, ntr = std::max(0, tr1 - tr0), sv = (int)sizeof(val_t);
    std::vector<RowCount> rc_((size_t)ntr);
    parallel_chunks(ntr, 1024, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) rc_[i] = count_row(T, tr0 + (int)i, tile_rowlen(tr0 + (int)i, tilem, rowA), tilen, colA, coo_in_tile, dense_mfma, csr_split, K.coo_cost);
    });
    std::vector<long long> pu((size_t)ntr + 1, 0), pc((size_t)ntr + 1, 0), ph((size_t)ntr + 1, 0), phv((size_t)ntr + 1, 0), phi((size_t)ntr + 1, 0);
    std::vector<long long> pd((size_t)ntr + 1, 0);
    for (int i = 0; i < ntr; i++) pd[i + 1] = pd[i] + rc_[i].ndense;
    const long long ND = pd[ntr];
    std::vector<DenseRow> drows;
    for (int i = 0; i < ntr; i++) {
        pu[i + 1] = pu[i] + rc_[i].nunits; pc[i + 1] = pc[i] + rc_[i].ncoo; ph[i + 1] = ph[i] + rc_[i].nheavy;
        phv[i + 1] = phv[i] + rc_[i].hval; phi[i + 1] = phi[i] + rc_[i].hidx;
    }
    const long long NU = pu[ntr], NC = pc[ntr], NH = ph[ntr], NHV = phv[ntr], NHI = phi[ntr];
    if (tilen > (1 << UNIT_FLAG_SHIFT)) { fprintf(stderr, "tilespmv: more than 2^24 column blocks: use TILESPMV_KERNEL=1\n"); return -2; }
    if (NU > INT32_MAX || NC > INT32_MAX || NH > INT32_MAX) { fprintf(stderr, "tilespmv: shard too large for 32-bit unit ids\n"); return -2; }

    // ---- strips (<= STRIP_MAX_ROWS whole tile-rows up to the cost target) for the unit kernel, one
    // heavy task per tile-row that owns heavy tiles, and pieces of very long tile-rows (all three
    // kinds of pieces write partial[] slots that k_fixup_split adds up in a fixed order).
    std::vector<STask> tasks;
    std::vector<Task> htasks;
    std::vector<FixRow> ifix, fix_late;   // split rows summed inside the unit kernel / by k_fixup_split after all passes
    const bool fix_inline_on = K.fix_inline != 0;
    // How the COO entry lists run (TILESPMV_WAVE_COO = 0 / 1 / 2 overrides):
    //   0  per 16-lane strip — regular matrices (a handful of entries per strip);
    //   1  per wavefront, the four strips' lists merged and ordered by column — entry-heavy but small grids, where the
    //      kernel is a chain of round trips and a workgroup barrier costs more than shared x lines save;
    //   2  per workgroup, the sixteen strips' lists merged and ordered by column — entry-heavy shards that fill the chip:
    //      distinct x lines per batch drop 3x and the CU's L1 -> L2 request rate is what bounds those.
    // Strip size.  Regular matrices: ~400 cost units (20 units) amortise the per-strip round trips; flat between 200 and 800
    // on large matrices.  Entry-heavy shards want MANY tile-rows per workgroup (power-law 8 M rows: 0.149 ms at 400,
    // 0.120 ms at 1600) but still about 3 workgroups per CU on small matrices (webbase-like: 12.9-13.1 us at ~760
    // workgroups, 13.6-14.3 at 1000, 15.4 at 570; scircuit-like flat 7.2-7.8 us from 250 to 670 workgroups).
    const bool entry_heavy = NC >= 5LL * ntr;   // (5 since round 3: an unaligned 7-point grid — 6 one-entry COO tiles per tile-row — runs 5 % faster with the workgroup entry mode; 4 per tile-row, the unaligned 5-point grid, does not)
    long long total_cost = 0;
    for (int i = 0; i < ntr; i++) total_cost += rc_[i].cost;
    // tilespmv_plan_spmm: k_units_mv walks a strip's entries with its 16 lanes, tile-row by tile-row; where entries are most
    // of the work (webbase-like: nvec 2 took 0.11 ms against 0.013 ms for one SpMV) one SpMV per right-hand side is faster
    const bool entry_dominated = (long long)K.coo_cost * NC * 2 > total_cost;
    int target = target_in;
    if (target <= 0) {
        target = 400;
        if (entry_heavy) {
            // balanced, entry-dominated shards (uniform random: 0.075 ms at 1600, 0.060 ms at 3200; band + random fill 0.126 -> 0.117)
            // take strips of up to 3200; skewed ones (R-MAT scale 20: 0.054 ms at 1600, 0.076 ms at 3200) and unit-dominated ones
            // (KKT-like 64^3: 0.022 ms at 967, 0.051 ms at 3200) stop at 1600
            long long max_cost = 0;
            for (int i = 0; i < ntr; i++) max_cost = std::max(max_cost, rc_[i].cost);
            const bool balanced = ntr > 0 && max_cost * ntr <= 4 * total_cost;
            const long long cap = (entry_dominated && balanced) ? 3200 : 1600;
            target = (int)std::min<long long>(cap, std::max<long long>(400, total_cost / (3 * 256 * 16)));
        } else if (total_cost / (16LL * 800) >= 4096 && NC <= 2LL * ntr) {
            // large regular shards: strips of up to 8 tile-rows once that still leaves >= 4096 workgroups (config 4: 0.1644-0.1665 -> 0.1606-0.1608 ms with the
            // nontemporal value stream, 5-pt 2896^2 0.0864 -> 0.0854; a 1024^2 grid would lose 17 % — 512 workgroups — and keeps 400) — and only while the
            // 8 tile-rows bring at most the 16 entries that travel with the unit prologue (4 entries per tile-row, the 4095^2 grid: 0.1875 ms at 800, 0.1770 at 400)
            target = 800;
        }
    }
    target = std::max(32, target);
    const long long est_wgs = total_cost / (16LL * target) + 1;
    const int wave_coo_env = K.entry_mode;
    const int entry_mode = wave_coo_env >= 0 ? std::min(2, wave_coo_env) : (!entry_heavy ? 0 : est_wgs < 768 ? 1 : 2);
    const bool wave_coo = entry_mode != 0;
    plan->entry_mode = entry_mode;
    // strips per workgroup: 32 (512 threads) only on request and only with the workgroup entry mode — twice as many tile-rows share
    // one column-ordered list (power-law 8 M rows: 0.204 -> 0.152 distinct 128-B x lines per entry) at the same 6 waves per SIMD, but
    // it measures slower everywhere (power-law 8 M 0.1038 -> 0.1072 ms, webbase-like 13.1 -> 13.9 us, KKT fp64 equal): default 16
    const int wg_strips = (entry_mode == 2 && K.wg_strips == 32) ? 32 : 16;
    plan->wg_strips = wg_strips;
    // Workgroup mode: the four wavefronts add into shared slabs.  Taking turns (4 barriers per trip) fixes the order of the
    // additions -> bit-reproducible sums; free or a gain on large grids (fewer LDS conflicts: power-law 8 M rows 0.122 ->
    // 0.118 ms), +8 % on mid-size ones (webbase-like 14.2 -> 15.3 us), which therefore add unordered unless
    // TILESPMV_COO_ORDERED=1 asks for reproducible bits.  Modes 0 and 1 are always ordered (one wavefront per slab).
    const int ordered_env = K.entry_ordered;
    const bool coo_ordered = ordered_env >= 0 ? ordered_env != 0 : est_wgs >= 2048;
    // ---- brick order (stencil-like shards): the grid strides of the shard are detected from its tile pattern, strips stay inside
    // one grid line, and after the cut the strips are regrouped so that the 16 strips of a workgroup — and the neighbouring
    // workgroups of an XCD window — form a brick of the grid instead of a run of one grid line: the x segments a tile-row shares
    // with its neighbours in the other two directions are then wanted at about the same time by one CU / one XCD, and hit in L1 /
    // L2 instead of being fetched again (nlpkkt160 stand-in: 3.07 -> 2.6-2.7 GB per launch at the fabric in fp64, 1.81 -> 1.56 GB in fp32;
    // time -2.5 ... -6 % in fp32, inside the matrix's 10 % placement spread in fp64: DESIGN.md S6.9).
    //   x_window  -1 (default): brick order on large 3-D shards   0: off   2: brick order wherever strides are found
    //              1: brick order + the workgroup's x segments staged once in LDS ("x windows": strips of at most XWIN_STRIP_ROWS
    //                 tile-rows; cuts another ~0.4 GB but runs 25 % slower — profiles/r03_xwindow_and_map.txt; opt-in only)
    int xs1 = K.x_stride1 > 0 ? K.x_stride1 : 0, xs2 = K.x_stride2 > 0 ? K.x_stride2 : 0;
    bool brick = K.x_window != 0 && wg_strips == 16 && (K.x_window > 0 || est_wgs >= 2048);
    if (brick && xs1 == 0) detect_strides(T, tr0, tr1, csr_split, dense_mfma, &xs1, &xs2);
    if (xs1 < 2 || (K.x_window < 0 && xs2 == 0)) brick = false;   // (2-D grids: measured neutral on the 5-point 4096^2 case)
    bool xwin = brick && K.x_window == 1 && entry_mode != 1;   // (the windowed kernel exists for entry modes 0 and 2)
    // (strips of at most 4 tile-rows in brick plans: nlpkkt160 stand-in fp64 0.418 -> 0.414 ms, fp32 0.252 -> 0.250 in one process; 2 rows: KKT 0.408 but 7-pt 256^3 +5 %)
    const int max_strip_rows = xwin ? XWIN_STRIP_ROWS : brick ? (K.brick_rows > 0 ? std::min(K.brick_rows, STRIP_MAX_ROWS) : 4) : STRIP_MAX_ROWS;
    if (brick && !K.xcd_from_caller) plan->xcd_chunk = 8;   // bricks are compact: smaller XCD windows keep an XCD's resident set together
    std::vector<unsigned char> row_k((size_t)ntr, 0), row_split((size_t)ntr, 0);
    const int npartial0 = npartial;
    auto cut = [&](int target) {
        // rows above this cost are cut into pieces.  With the wavefront / workgroup entry modes a long row is no longer one strip's
        // private burden, but an unsplit one still makes its workgroup the last to finish: the threshold stops growing with the
        // strip size there (R-MAT scale 20 at strip size 3200: 0.099 ms with rows of up to 19,200 cost units kept whole)
        const int split_cap = K.split_cap;
        const int split_above = wave_coo ? std::max(split_above_in, std::min(6 * target, split_cap)) : std::max(6 * target, split_above_in);
        const int piece = std::max(wave_coo ? std::min(2 * target, 1600) : 2 * target, split_above / 3);
        tasks.clear(); htasks.clear(); ifix.clear(); fix_late.clear(); fix.clear(); drows.clear(); npartial = npartial0;
        std::fill(row_k.begin(), row_k.end(), 0); std::fill(row_split.begin(), row_split.end(), 0);
        const int strip_even = K.strip_even;  // 0 off, 1 = value group, n > 1 = multiples of n units
        auto blank = [&]() { STask k; memset(&k, 0, sizeof(k)); k.partial = -1; return k; };
        auto is_heavy = [&](int t) {
            const int fmt = T->Format[t];
            return fmt == TILESPMV_FMT_CSR && !csr_split;
        };
        auto heavy_sizes = [&](int t, int *nv, int *ni) {
            const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t];
            (void)fmt; *nv = stored; *ni = 16 + (stored + 1) / 2;
        };
        // k_dense_mfma broadcasts the column blocks of one DenseRow piece from a single 64-lane load: a piece holds at most
        // 64 dense tiles.  Split rows cut their dense tiles into pieces of 32; an unsplit row is one piece, so a row with
        // more dense tiles than that is always split, whatever the cost knobs say (TILESPMV_STRIP_COST / _SPLIT_ABOVE).
        constexpr int DENSE_PIECE = 32;
        auto must_split = [&](int i) { return rc_[i].cost > split_above || rc_[i].ndense > DENSE_PIECE; };
        for (int i = 0; i < ntr;) {
            if (must_split(i)) {
                row_split[i] = 1;
                FixRow f{tr0 + i, npartial, 0, 0};
                // entry pieces: four consecutive pieces share a wavefront, which walks their lists together (4 x 192 = 2 trips of 6 x 64)
                const int pu_ = std::max(1, piece / 16), pc_ = std::max(16, K.coo_piece > 0 ? K.coo_piece : (entry_mode == 1 ? 192 : piece / std::max(1, K.coo_cost)));
                for (long long u = pu[i]; u < pu[i + 1]; u += pu_) {
                    STask k = blank(); k.row = tr0 + i; k.nrows = 1; k.partial = npartial++;
                    k.unit_begin = (int)u; k.unit_end = (int)std::min(pu[i + 1], u + pu_);
                    tasks.push_back(k); f.count++;
                }
                for (long long c = pc[i]; c < pc[i + 1]; c += pc_) {
                    STask k = blank(); k.row = tr0 + i; k.nrows = 1; k.partial = npartial++;
                    k.coo_begin = (int)c; k.coo_end = (int)std::min(pc[i + 1], c + pc_);
                    tasks.push_back(k); f.count++;
                }
                const int stream_pieces = f.count;  // pieces executed by the unit kernel
                long long h = ph[i], hv = phv[i], hi = phi[i];
                int t = T->tile_ptr[tr0 + i];
                while (h < ph[i + 1]) {  // heavy tiles of a split row: cut at tile boundaries by payload size
                    Task k{(int)h, (int)h, hv, hi, tr0 + i, npartial++};
                    long long c = 0;
                    while (h < ph[i + 1] && (c == 0 || c < piece)) {
                        while (!is_heavy(t)) t++;
                        int nv, ni; heavy_sizes(t, &nv, &ni);
                        hv += nv; hi += ni; c += nv + 256; h++; t++;
                    }
                    k.tile_end = (int)h;
                    htasks.push_back(k); f.count++;
                }
                for (long long dq = pd[i]; dq < pd[i + 1]; dq += DENSE_PIECE) {  // dense tiles of a split row: 32 per piece
                    drows.push_back(DenseRow{tr0 + i, (int)dq, (int)std::min(pd[i + 1], dq + DENSE_PIECE), npartial++});
                    f.count++;
                }
                // all pieces inside the unit kernel -> the last one to finish adds the slots up there
                const bool inline_fix = fix_inline_on && f.count == stream_pieces;
                for (int q = 0; q < stream_pieces; q++) tasks[tasks.size() - 1 - (size_t)q].nounit_mask = inline_fix ? (unsigned)ifix.size() : 0xFFFFFFFFu;
                if (inline_fix) ifix.push_back(f); else fix_late.push_back(f);
                fix.push_back(f);
                i++;
                continue;
            }
            STask k = blank();
            k.row = tr0 + i;
            k.unit_begin = (int)pu[i]; k.coo_begin = (int)pc[i];
            long long c = 0;
            int j = i;
            // how many tile-rows: up to the cost target, then nudged by one row either way if that leaves fewer padding
            // units (the strip's values are stored in groups of UNIT_GROUP units, tail padded with zero units)
            int jend = i;
            {
                long long cc = 0;
                while (jend < ntr && jend - i < max_strip_rows && !must_split(jend)) {
                    if (brick && jend > i && (tr0 + jend) % xs1 == 0) break;   // brick order: a strip stays inside one grid line
                    const long long nc = cc + rc_[jend].cost;
                    // entry-heavy shards round to the nearest strip size (rows cost 100-400 each there: "never above the target"
                    // would leave most strips half empty and double the number of wavefronts)
                    if (jend > i && nc > target && !(wave_coo && nc - target < target - cc && nc <= target + target / 2)) break;
                    cc = nc; jend++;
                }
                // (whole batches of 4 units, which are also whole value groups: a half-empty last batch costs as much as a full one)
                const int quantum = strip_even > 1 ? strip_even : UNIT_GROUP;
                auto pad = [&](int e) { return (int)((quantum - (pu[e] - pu[i]) % quantum) % quantum); };
                if (strip_even && pad(jend) > 0) {
                    int best = jend;
                    if (jend < ntr && jend - i < max_strip_rows && !(brick && (tr0 + jend) % xs1 == 0) && !must_split(jend) && cc + rc_[jend].cost <= target + target / 3 && pad(jend + 1) < pad(best)) best = jend + 1;
                    if (best == jend && jend - i >= 3 && pad(jend - 1) < pad(best)) best = jend - 1;
                    jend = best;
                }
            }
            while (j < jend) {
                row_k[j] = (unsigned char)(j - i);
                if (rc_[j].nunits == 0) k.nounit_mask |= 1u << (j - i);
                if (rc_[j].nheavy > 0) htasks.push_back(Task{(int)ph[j], (int)ph[j + 1], phv[j], phi[j], tr0 + j, -1});
                if (rc_[j].ndense > 0) drows.push_back(DenseRow{tr0 + j, (int)pd[j], (int)pd[j + 1], -1});
                c += rc_[j].cost; j++;
            }
            k.nrows = j - i;
            k.unit_end = (int)pu[j]; k.coo_end = (int)pc[j];
            tasks.push_back(k);
            i = j;
        }

    };
    cut(target);
    // ---- fill
    std::vector<uint4> h_udesc((size_t)NU);
    val_t *h_uval = zalloc<val_t>((size_t)NU * 16);
    val_t *h_cval = zalloc<val_t>((size_t)NC);
    std::vector<int> h_ccol((size_t)NC);
    std::vector<unsigned char> h_crow((size_t)NC);
    std::vector<uint2> h_hdesc((size_t)NH);
    val_t *h_hval = zalloc<val_t>((size_t)NHV);
    unsigned char *h_hidx = zalloc<unsigned char>((size_t)NHI + 16);
    std::vector<int> h_dcb((size_t)ND);
    val_t *h_dval = zalloc<val_t>((size_t)ND * 256);
    parallel_chunks(ntr, 256, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            const int bi = tr0 + (int)i, rowlen = tile_rowlen(bi, tilem, rowA);
            const unsigned kr = row_k[i];
            long long u = pu[i], c = pc[i], h = ph[i], hv = phv[i], hi = phi[i], dq = pd[i];
            auto put_unit = [&](int cb, const val_t *src, int stride_ok_rows, unsigned long long nibs) {
                // src: rowlen consecutive values of this column; nibs: 16 nibbles, row 0 in the top nibble
                for (int r = 0; r < stride_ok_rows; r++) h_uval[u * 16 + r] = src[r];
                const unsigned w0 = (unsigned)cb | ((kr << UNIT_ROW_SHIFT) << UNIT_FLAG_SHIFT);
                h_udesc[(size_t)u] = make_uint4(w0, (unsigned)(nibs >> 32), w0, (unsigned)(nibs & 0xffffffffull));
                u++;
            };
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                const int fmt = T->Format[t], stored = T->blknnz[t + 1] - T->blknnz[t], w = T->tilewidth[t];
                const int cb = T->tile_columnidx[t], collen = tile_collen(cb, tilen, colA);
                switch (fmt) {
                case TILESPMV_FMT_ELL: {
                    const int off = T->ell_offset[t];
                    for (int s = 0; s < w; s++) {
                        unsigned long long nibs = 0;
                        for (int r = 0; r < rowlen; r++) nibs |= (unsigned long long)nib(T->ell_compressedIdx, (long long)off + s * rowlen + r) << (60 - 4 * r);
                        put_unit(cb, T->Blockell_Val + off + s * rowlen, rowlen, nibs);
                    }
                    break;
                }
                case TILESPMV_FMT_HYB: {
                    const int off = T->hyb_offset[t], nell = w * rowlen;
                    const unsigned char *src = T->hybIdx + hyb_off[t];
                    for (int s = 0; s < w; s++) {
                        unsigned long long nibs = 0;
                        for (int r = 0; r < rowlen; r++) nibs |= (unsigned long long)nib(src, s * rowlen + r) << (60 - 4 * r);
                        put_unit(cb, T->Blockhyb_Val + off + s * rowlen, rowlen, nibs);
                    }
                    if (coo_in_tile)
                        for (int q = 0; q < stored - nell; q++) {
                            const unsigned char rcb = src[(nell + 1) / 2 + q];
                            h_cval[c] = T->Blockhyb_Val[off + nell + q]; h_ccol[(size_t)c] = cb * 16 + (rcb & 15);
                            h_crow[(size_t)c] = (unsigned char)((kr << 4) | (rcb >> 4)); c++;
                        }
                    break;
                }
                case TILESPMV_FMT_DNSCOL: {
                    const int off = T->dnscol_offset[t], co = T->dnscolptr[t], k = T->dnscolptr[t + 1] - co;
                    for (int q = 0; q < k; q++) put_unit(cb, T->Blockdensecol_Val + off + q * rowlen, rowlen, 0x1111111111111111ull * (unsigned)(T->densecolid[co + q] & 15));
                    break;
                }
                case TILESPMV_FMT_COO:
                    if (coo_in_tile) {
                        const int off = T->coo_offset[t];
                        for (int q = 0; q < stored; q++) {
                            const unsigned char rcb = T->coo_compressed_Idx[off + q];
                            h_cval[c] = T->Blockcoo_Val[off + q]; h_ccol[(size_t)c] = cb * 16 + (rcb & 15);
                            h_crow[(size_t)c] = (unsigned char)((kr << 4) | (rcb >> 4)); c++;
                        }
                    }
                    break;
                case TILESPMV_FMT_DNS:
                    if (!dense_mfma) {
                        const int off = T->dns_offset[t];
                        for (int q = 0; q < collen; q++) put_unit(cb, T->Blockdense_Val + off + q * rowlen, rowlen, 0x1111111111111111ull * (unsigned)q);
                        break;
                    }
                    {   // dense tile for the matrix cores: 256 values, zero padded, in MFMA operand order (dense_slot, hip_plan.h)
                        const int off = T->dns_offset[t];
                        val_t *dst = h_dval + dq * 256;
                        for (int cc = 0; cc < collen; cc++)
                            for (int r = 0; r < rowlen; r++) dst[dense_slot(r, cc)] = T->Blockdense_Val[off + cc * rowlen + r];
                        h_dcb[(size_t)dq] = cb;
                        dq++;
                    }
                    break;
                case TILESPMV_FMT_DNSROW: {
                    const int off = T->dnsrow_offset[t], ro = T->dnsrowptr[t], k = T->dnsrowptr[t + 1] - ro;
                    for (int q = 0; q < k; q++) {
                        for (int cc = 0; cc < collen; cc++) h_uval[u * 16 + cc] = T->Blockdenserow_Val[off + q * collen + cc];
                        const unsigned w0 = (unsigned)cb | (((kr << UNIT_ROW_SHIFT) | UNIT_ROWUNIT) << UNIT_FLAG_SHIFT);
                        const unsigned rid = (unsigned)(T->denserowid[ro + q] & 15);
                        h_udesc[(size_t)u] = make_uint4(w0, rid, w0, rid);
                        u++;
                    }
                    break;
                }
                case TILESPMV_FMT_CSR:
                    if (csr_split) {
                        const int off = T->csr_offset[t];
                        const unsigned char *ptr = T->Blockcsr_Ptr + T->csrptr_offset[t];
                        int rem;
                        const int w = csr_split_width(ptr, rowlen, stored, &rem);
                        const long long u0 = u;
                        for (int sidx = 0; sidx < w; sidx++) {  // descriptors first (zero nibbles), payload below
                            const unsigned w0 = (unsigned)cb | ((kr << UNIT_ROW_SHIFT) << UNIT_FLAG_SHIFT);
                            h_udesc[(size_t)u] = make_uint4(w0, 0u, w0, 0u);
                            u++;
                        }
                        for (int r = 0; r < rowlen; r++) {
                            const int k0 = ptr[r], k1 = (r == rowlen - 1) ? stored : ptr[r + 1];
                            for (int kk = k0; kk < k1; kk++) {
                                const int lc = nib(T->csr_compressedIdx, (long long)off + kk), sidx = kk - k0;
                                if (sidx < w) {
                                    h_uval[(u0 + sidx) * 16 + r] = T->Blockcsr_Val[off + kk];
                                    if (r < 8) h_udesc[(size_t)(u0 + sidx)].y |= (unsigned)lc << (28 - 4 * r);
                                    else h_udesc[(size_t)(u0 + sidx)].w |= (unsigned)lc << (28 - 4 * (r - 8));
                                } else {
                                    h_cval[c] = T->Blockcsr_Val[off + kk]; h_ccol[(size_t)c] = cb * 16 + lc;
                                    h_crow[(size_t)c] = (unsigned char)((kr << 4) | r); c++;
                                }
                            }
                        }
                        break;
                    }
                    // fallthrough: CSR tile as a heavy (whole) tile
                {
                    Emit em = emit_of(T, t, rowlen, true);
                    repack_tile(T, t, em, rowlen, collen, 0, h_hval + hv, h_hidx + hi);
                    h_hdesc[(size_t)h] = make_uint2((unsigned)cb, (unsigned)em.fmt | ((unsigned)em.p1 << DESC_P1_SHIFT));
                    h++; hv += em.nv; hi += em.ni;
                    break;
                }
                }
            }
            if (!row_split[i] && u > pu[i]) { h_udesc[(size_t)u - 1].x |= UNIT_EOR << UNIT_FLAG_SHIFT; h_udesc[(size_t)u - 1].z |= UNIT_EOR << UNIT_FLAG_SHIFT; }
            if (h > ph[i]) h_hdesc[(size_t)h - 1].y |= DESC_EOR;
        }
    });

    int rc = 0;
    DevStream &S = plan->st;
    // ---- x windows: brick order of the strips, then one window of column blocks per workgroup
    std::vector<uint4> h_udesc_cb;   // the descriptors with column blocks (multi-vector kernel), when windows put slots into h_udesc
    std::vector<int2> h_wg_win;
    std::vector<int> h_win_cb;
    int xwin_slots_max = 0;
    long long xwin_segments = 0, xwin_wgs = 0;
    if (brick && !tasks.empty()) {
        const size_t nt = tasks.size();
        // grid coordinates of every strip: position in its line (ordinal of the strip), line in its plane, plane
        std::vector<int> sx(nt), ly(nt), lz(nt);
        {
            long long prev_line = -1; int ord = 0;
            for (size_t i = 0; i < nt; i++) {
                const long long line = tasks[i].row / xs1;
                ord = line == prev_line ? ord + 1 : 0;
                prev_line = line;
                sx[i] = ord;
                ly[i] = xs2 ? (int)(line % (xs2 / xs1)) : (int)line;
                lz[i] = xs2 ? tasks[i].row / xs2 : 0;
            }
        }
        auto blocks_of = [&](const STask &k, std::vector<int> &out) {
            for (int u = k.unit_begin; u < k.unit_end; u++) out.push_back((int)(h_udesc[(size_t)u].x & 0xFFFFFFu));
        };
        struct Shape { int px, py, pz; };
        const Shape shapes3[] = {{1, 4, 4}, {2, 2, 4}, {2, 4, 2}, {4, 2, 2}, {1, 2, 8}, {1, 8, 2}, {4, 4, 1}, {2, 8, 1}, {1, 16, 1}, {16, 1, 1}};
        const Shape shapes2[] = {{4, 4, 1}, {2, 8, 1}, {8, 2, 1}, {1, 16, 1}, {16, 1, 1}};
        const Shape *shapes = xs2 ? shapes3 : shapes2;
        const int nshapes = xs2 ? 10 : 5;
        std::vector<unsigned> order(nt), best_order;
        double best_avg = 1e30;
        Shape best_shape{16, 1, 1};
        auto sort_for = [&](const Shape &sh) {
            for (size_t i = 0; i < nt; i++) order[i] = (unsigned)i;
            std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
                const int ka[6] = {lz[a] / sh.pz, ly[a] / sh.py, sx[a] / sh.px, lz[a] % sh.pz, ly[a] % sh.py, sx[a] % sh.px};
                const int kb[6] = {lz[b] / sh.pz, ly[b] / sh.py, sx[b] / sh.px, lz[b] % sh.pz, ly[b] % sh.py, sx[b] % sh.px};
                for (int q = 0; q < 6; q++) if (ka[q] != kb[q]) return ka[q] < kb[q];
                return a < b;
            });
        };
        const size_t nwg = (nt + 15) / 16;
        std::vector<int> tmp;
        for (int si = 0; si < nshapes; si++) {   // the brick shape that needs the fewest window slots on a sample of workgroups
            sort_for(shapes[si]);
            long long slots = 0, wgs = 0;
            for (size_t w = nwg / 128; w < nwg; w += std::max<size_t>(1, nwg / 64)) {
                tmp.clear();
                for (size_t t = 16 * w; t < std::min(nt, 16 * w + 16); t++) blocks_of(tasks[order[t]], tmp);
                std::sort(tmp.begin(), tmp.end());
                slots += (long long)(std::unique(tmp.begin(), tmp.end()) - tmp.begin()); wgs++;
            }
            const double avg = wgs ? (double)slots / (double)wgs : 1e30;
            if (avg < best_avg * 0.98) { best_avg = avg; best_order = order; best_shape = shapes[si]; }
        }
        {
            std::vector<STask> permuted(nt);
            for (size_t i = 0; i < nt; i++) permuted[i] = tasks[best_order[i]];
            tasks.swap(permuted);
        }
        h_udesc_cb = h_udesc;
        h_wg_win.assign(nwg, make_int2(0, 0));
        std::vector<std::vector<int>> wg_blocks(nwg);
        const bool order_only = !xwin;   // brick order alone: x is still gathered from global memory (through L1 / L2)
        parallel_chunks(order_only ? 0 : (int64_t)nwg, 64, [&](int64_t b, int64_t e, int) {
            for (int64_t w = b; w < e; w++) {
                std::vector<int> &bl = wg_blocks[(size_t)w];
                for (size_t t = 16 * (size_t)w; t < std::min(nt, 16 * (size_t)w + 16); t++) blocks_of(tasks[t], bl);
                std::sort(bl.begin(), bl.end());
                bl.erase(std::unique(bl.begin(), bl.end()), bl.end());
                if (bl.size() > (size_t)XWIN_MAX_SLOTS) { bl.clear(); continue; }   // this workgroup reads x from global memory
                for (size_t t = 16 * (size_t)w; t < std::min(nt, 16 * (size_t)w + 16); t++)
                    for (int u = tasks[t].unit_begin; u < tasks[t].unit_end; u++) {
                        uint4 &d = h_udesc[(size_t)u];
                        const unsigned slot = (unsigned)(std::lower_bound(bl.begin(), bl.end(), (int)(d.x & 0xFFFFFFu)) - bl.begin());
                        d.x = (d.x & 0xFF000000u) | slot; d.z = d.x;
                    }
            }
        });
        for (size_t w = 0; w < nwg; w++) {
            h_wg_win[w] = make_int2((int)h_win_cb.size(), (int)wg_blocks[w].size());
            h_win_cb.insert(h_win_cb.end(), wg_blocks[w].begin(), wg_blocks[w].end());
            xwin_slots_max = std::max(xwin_slots_max, (int)wg_blocks[w].size());
            xwin_segments += (long long)wg_blocks[w].size(); xwin_wgs += !wg_blocks[w].empty();
        }
        if (xwin_slots_max == 0) { xwin = false; h_udesc_cb.clear(); }
        if (getenv("TILESPMV_PLAN_VERBOSE"))
            fprintf(stderr, "tilespmv: brick order: strides %d / %d tile-rows, brick %d x %d x %d strips, %.1f distinct column blocks per workgroup on the sample; x windows: %lld of %zu workgroups, %d slots at most\n",
                    xs1, xs2, best_shape.px, best_shape.py, best_shape.pz, best_avg, xwin_wgs, nwg, xwin_slots_max);
    } else { xwin = false; brick = false; }
    plan->info[TILESPMV_INFO_BRICK_ORDER] = brick ? 1 : 0;
    plan->xwin_lds_bytes = xwin ? xwin_slots_max * 16 * (int)sizeof(val_t) : 0;
    plan->size_hint = (size_t)(NU * (12 + 16LL * sv) + NC * (2LL * sv + 13) + NHV * sv + NHI + ND * (4 + 256LL * sv) + (long long)tasks.size() * 40);   // estimate of the plan's bytes: picks the block size of upload()
    // ---- final HBM form of the unit streams.  Descriptors: 12 B (the duplicate of word 0 is dropped).  Values: the
    // units of one task are stored in GROUPS of G = 16 / sizeof(value) units (2 in fp64, 4 in fp32) — the values of
    // the G units interleaved per row, so that a lane fetches G units with one 16-byte load (row r of the group at
    // +16 r bytes).  A task whose unit count is not a multiple of G gets padding units (zero values, never executed:
    // unit_end excludes them) so that its last group exists.
    constexpr long long G = UNIT_GROUP;
    auto padded = [&](long long n) { return (n + G - 1) / G * G; };
    long long NUP = 0;
    for (const STask &k : tasks) NUP += padded(k.unit_end - k.unit_begin);
    if (NUP > INT32_MAX) {
        fprintf(stderr, "tilespmv: shard too large for 32-bit unit ids\n");
        free(h_uval); free(h_cval); free(h_hval); free(h_hidx); free(h_dval);
        return -2;
    }
    {
        std::vector<UDesc> packed((size_t)NUP, UDesc{0u, 0u, 0u});
        val_t *paired = zalloc<val_t>((size_t)NUP * 16);
        std::vector<long long> new_begin(tasks.size()), old_begin(tasks.size());
        for (size_t i = 0; i < tasks.size(); i++) old_begin[i] = tasks[i].unit_begin;
        long long at = 0;
        for (size_t i = 0; i < tasks.size(); i++) { new_begin[i] = at; at += padded(tasks[i].unit_end - tasks[i].unit_begin); }
        parallel_chunks((int64_t)tasks.size(), 512, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; i++) {
                STask &k = tasks[(size_t)i];
                const long long ub = k.unit_begin, n = k.unit_end - ub, nb = new_begin[(size_t)i];
                for (long long j = 0; j < n; j++) {
                    const uint4 d = h_udesc[(size_t)(ub + j)];
                    packed[(size_t)(nb + j)] = UDesc{d.x, d.y, d.w};
                    const val_t *src = h_uval + (ub + j) * 16;
                    val_t *dst = paired + (nb + j / G * G) * 16 + (j % G);
                    for (int r = 0; r < 16; r++) dst[G * r] = src[r];
                }
                if (n > 0) { k.unit_begin = (int)nb; k.unit_end = (int)(nb + n); }
            }
        });
        // ---- 4-B descriptors where the units of the shard use few distinct column patterns (stencil-like shards: 4 patterns in the
        // 5- and 7-point grids, 36 in the KKT stand-in): column block | pattern id << cb_bits | flags << 27, the patterns (the
        // two nibble words) in a dictionary the kernels gather from.  Not for x-window plans (their descriptors hold slots).
        S.udict = nullptr; S.cb_bits = 0;
        std::vector<uint2> dict;
        std::vector<unsigned> compact;
        // ... and only where it pays: 8 bytes per unit must be at least 2 % of the streams (an entry-dominated plan with a handful of units would only buy the dictionary
        // hop at the start of every strip: webbase-1M stand-in 13.2 -> 13.6 us); desc_dict = 1 asks for it wherever it is possible
        const bool dict_pays = K.desc_dict > 0 ? true : 8LL * NUP * 50 >= NUP * (12 + 16LL * sv) + NC * (sv + 4LL);
        if (K.desc_dict != 0 && dict_pays && !xwin && NUP > 0) {
            const int cb_bits = std::max(1, 32 - __builtin_clz((unsigned)std::max(1, T->tilen - 1)));
            const int pid_bits = std::min(DICT_MAX_BITS, 27 - cb_bits);
            if (pid_bits >= 1) {
                const size_t cap = (size_t)1 << pid_bits;
                std::vector<std::unordered_set<unsigned long long>> local((size_t)host_threads());
                std::atomic<int> over(0);
                parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int th) {
                    if (over.load(std::memory_order_relaxed)) return;
                    std::unordered_set<unsigned long long> &L = local[(size_t)th];
                    for (int64_t u = b; u < e; u++) {
                        L.insert(((unsigned long long)packed[(size_t)u].n0 << 32) | packed[(size_t)u].n1);
                        if (L.size() > cap) { over.store(1); return; }
                    }
                });
                std::vector<unsigned long long> all;
                if (!over.load()) {
                    for (auto &L : local) all.insert(all.end(), L.begin(), L.end());
                    std::sort(all.begin(), all.end());
                    all.erase(std::unique(all.begin(), all.end()), all.end());
                }
                if (!over.load() && all.size() <= cap) {
                    dict.resize(all.size());
                    for (size_t i = 0; i < all.size(); i++) dict[i] = make_uint2((unsigned)(all[i] >> 32), (unsigned)(all[i] & 0xffffffffull));
                    compact.resize((size_t)NUP);
                    const unsigned cbmask = (1u << cb_bits) - 1u;
                    parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int) {
                        for (int64_t u = b; u < e; u++) {
                            const UDesc &d = packed[(size_t)u];
                            const unsigned long long key = ((unsigned long long)d.n0 << 32) | d.n1;
                            const unsigned pid = (unsigned)(std::lower_bound(all.begin(), all.end(), key) - all.begin());
                            compact[(size_t)u] = (d.w0 & cbmask) | (pid << cb_bits) | ((d.w0 >> UNIT_FLAG_SHIFT) << 27);
                        }
                    });
                    S.cb_bits = cb_bits;
                }
            }
        }
        if (S.cb_bits > 0) {
            rc |= plan->upload(compact.data(), compact.size(), reinterpret_cast<const unsigned **>(&S.udesc));
            rc |= plan->upload(dict.data(), dict.size(), &S.udict);
        } else rc |= plan->upload(packed.data(), (size_t)NUP, &S.udesc);
        plan->info[TILESPMV_INFO_DESC_BYTES] = S.cb_bits > 0 ? 4 : 12;
        rc |= plan->upload(paired, (size_t)NUP * 16, &S.uval);
        free(paired);
        S.udesc_cb = S.udesc;
        if (xwin) {   // the multi-vector kernel keeps reading x from global memory: its descriptors carry column blocks
            std::fill(packed.begin(), packed.end(), UDesc{0u, 0u, 0u});
            parallel_chunks((int64_t)tasks.size(), 512, [&](int64_t b, int64_t e, int) {
                for (int64_t i = b; i < e; i++) {
                    const STask &k = tasks[(size_t)i];   // (unit_begin already points into the packed numbering)
                    for (long long j = 0; j < k.unit_end - k.unit_begin; j++) {
                        const uint4 d = h_udesc_cb[(size_t)(old_begin[(size_t)i] + j)];
                        packed[(size_t)(k.unit_begin + j)] = UDesc{d.x, d.y, d.w};
                    }
                }
            });
            rc |= plan->upload(packed.data(), (size_t)NUP, &S.udesc_cb);
            rc |= plan->upload(h_wg_win.data(), h_wg_win.size(), &S.wg_win);
            rc |= plan->upload(h_win_cb.data(), h_win_cb.size(), &S.win_cb);
        } else { S.wg_win = nullptr; S.win_cb = nullptr; }
    }
    S.wg_coo = nullptr; S.grec = nullptr; S.gbase = nullptr; S.dest_bits = 11;
    long long n_rec = 0, n_chunk = 0, n_groups = 0;
    if (entry_mode != 0) {
        const size_t GS = entry_mode == 2 ? (size_t)wg_strips : 4;   // tasks whose lists are merged: one workgroup's or one wavefront's
        const int slab_shift = xwin ? 6 : 7;   // a strip's slab of s_y: XWIN_STRIP_ROWS x 16 values in x-window plans, STRIP_MAX_ROWS x 16 otherwise
        const int dest_bits = entry_mode == 2 ? (wg_strips == 32 ? 12 : 4 + slab_shift) : 9;   // strip-in-group | row-in-strip | row (4)
        S.dest_bits = dest_bits;
        const size_t nwg = (tasks.size() + GS - 1) / GS;
        std::vector<std::vector<ERec>> grp_rec(nwg);
        std::vector<std::vector<unsigned>> grp_base(nwg);
        std::atomic<int> bad(0);
        parallel_chunks((int64_t)nwg, 64, [&](int64_t b, int64_t e, int) {
            std::vector<std::pair<unsigned long long, unsigned>> key;   // (column << 32 | position in strip / list order, destination)
            std::vector<int> src;
            std::vector<PEnt> ents;
            for (int64_t w = b; w < e; w++) {
                key.clear(); src.clear();
                for (size_t t = GS * (size_t)w; t < std::min(tasks.size(), GS * (size_t)w + GS); t++)
                    for (int q = tasks[t].coo_begin; q < tasks[t].coo_end; q++) {   // column-major order; ties keep strip / list order
                        key.push_back({((unsigned long long)(unsigned)h_ccol[(size_t)q] << 32) | (unsigned long long)key.size(),
                                       (unsigned)((t & (GS - 1)) << slab_shift) | (unsigned)h_crow[(size_t)q]});
                        src.push_back(q);
                    }
                std::sort(key.begin(), key.end());
                ents.resize(key.size());
                for (size_t i = 0; i < key.size(); i++) {
                    const int q = src[(size_t)(key[i].first & 0xFFFFFFFFull)];
                    ents[i] = PEnt{(unsigned)h_ccol[(size_t)q], key[i].second, h_cval[q]};
                }
                if (!pack_list(ents, dest_bits, grp_rec[(size_t)w], grp_base[(size_t)w], plan->dry)) bad++;
            }
        });
        if (bad.load()) { fprintf(stderr, "tilespmv: internal error: %d packed entry lists do not decode to their entries\n", bad.load()); rc = -6; }
        std::vector<int4> wg((size_t)nwg);
        for (size_t w = 0; w < nwg; w++) {
            wg[w] = make_int4((int)n_rec, (int)(n_rec + (long long)grp_rec[w].size()), (int)n_chunk, 0);
            n_rec += (long long)grp_rec[w].size(); n_chunk += (long long)grp_base[w].size();
        }
        if (n_rec > INT32_MAX) { fprintf(stderr, "tilespmv: shard too large for 32-bit entry ids\n"); rc = -2; n_rec = 0; }
        std::vector<ERec> g_rec((size_t)n_rec);
        std::vector<unsigned> g_base((size_t)n_chunk);
        if (rc == 0)
            parallel_chunks((int64_t)nwg, 256, [&](int64_t b, int64_t e, int) {
                for (int64_t w = b; w < e; w++) {
                    if (!grp_rec[(size_t)w].empty()) memcpy(&g_rec[(size_t)wg[(size_t)w].x], grp_rec[(size_t)w].data(), grp_rec[(size_t)w].size() * sizeof(ERec));
                    if (!grp_base[(size_t)w].empty()) memcpy(&g_base[(size_t)wg[(size_t)w].z], grp_base[(size_t)w].data(), grp_base[(size_t)w].size() * sizeof(unsigned));
                }
            });
        n_groups = (long long)nwg;
        rc |= plan->upload(wg.data(), wg.size(), &S.wg_coo);
        rc |= plan->upload(g_rec.data(), g_rec.size(), &S.grec);
        rc |= plan->upload(g_base.data(), g_base.size(), &S.gbase);
    }
    rc |= plan->upload(h_cval, (size_t)NC, &S.cval);
    rc |= plan->upload(h_ccol.data(), (size_t)NC, &S.ccol);
    rc |= plan->upload(h_crow.data(), (size_t)NC, &S.crow);
    DevPlan &D = plan->dev;  // heavy tiles reuse the first-generation streams + kernel (accumulate mode)
    rc |= plan->upload(h_hdesc.data(), (size_t)NH, &D.desc);
    rc |= plan->upload(h_hval, (size_t)NHV, &D.val);
    rc |= plan->upload(h_hidx, (size_t)NHI, &D.idx);
    rc |= plan->upload(htasks.data(), htasks.size(), &D.task);
    D.ntasks = (int)htasks.size();
    rc |= plan->upload(tasks.data(), tasks.size(), &S.task);
    rc |= plan->upload(h_dcb.data(), (size_t)ND, &plan->dn.cb);
    rc |= plan->upload(h_dval, (size_t)ND * 256, &plan->dn.val);
    rc |= plan->upload(drows.data(), drows.size(), &plan->dn.rows);
    plan->dn.nrows = (int)drows.size();
    for (const DenseRow &dr : drows)
        if (dr.tile_end - dr.tile_begin > 64) { fprintf(stderr, "tilespmv: internal error: dense piece of %d tiles\n", dr.tile_end - dr.tile_begin); rc = -6; }
    free(h_uval); free(h_cval); free(h_hval); free(h_hidx); free(h_dval);
    S.ntasks = (int)tasks.size();
#ifdef TILESPMV_STAMPS
    { void *sp = nullptr; const size_t nst = ((tasks.size() + 15) / 16) * 4 * 8;
      if (hipMalloc(&sp, nst * 8 + 64) == hipSuccess) { (void)hipMemset(sp, 0, nst * 8 + 64); plan->allocs.push_back(sp); } S.stamps = (unsigned long long *)sp; }
#endif
    S.ifix = nullptr; S.ifix_count = nullptr;
    if (!ifix.empty()) {
        rc |= plan->upload(ifix.data(), ifix.size(), &S.ifix);
        std::vector<unsigned> zeros(ifix.size(), 0u);
        const unsigned *cnt = nullptr;
        rc |= plan->upload(zeros.data(), zeros.size(), &cnt);
        S.ifix_count = const_cast<unsigned *>(cnt);
    }
    rc |= plan->upload(fix_late.data(), fix_late.size(), &plan->dev.fix_late);
    plan->dev.nfix_late = (int)fix_late.size();
    // entry mode 0 only: strips with more entries than this run their list before the unit pipeline (32: swept on KKT fp64 / scircuit /
    // webbase stand-ins in round 1, best or within 1 %)
    S.coo_heavy_min = K.coo_heavy_min;
    S.coo_ordered = coo_ordered ? 1 : 0;
    // y stores: streaming (nontemporal) where y is a real share of what the launch moves — they keep y from displacing x in L2: config 4 0.1946 -> 0.1845 ms,
    // 7-pt 256^3 0.2533 -> 0.2432, power-law 8 M 0.1078 -> 0.1043 — plain where it is a few per cent: there the streaming form buys nothing and makes the time
    // depend on where the CALLER's y happens to sit (nlpkkt160 stand-in fp64: 0.413 or 0.459 ms by the copy of y; plain: 0.408-0.411 with every copy)
    {
        const long long stream_b = NU * ((S.cb_bits > 0 ? 4 : 12) + 16LL * sv) + NC * (sv + 4LL), y_b = 16LL * ntr * sv;
        S.y_streaming = K.y_store >= 0 ? (K.y_store != 0) : (y_b * 20 >= stream_b);   // >= 5 %
    }
    plan->info[TILESPMV_INFO_ENTRY_MODE] = entry_mode;
    plan->info[TILESPMV_INFO_ENTRY_ORDERED] = (entry_mode != 2 || coo_ordered) ? 1 : 0;
    plan->info[TILESPMV_INFO_STRIP_COST] = target;
    plan->info[TILESPMV_INFO_WG_STRIPS] = wg_strips;
    plan->info[TILESPMV_INFO_X_WINDOW_SLOTS] = xwin ? xwin_slots_max : 0;
    plan->info[TILESPMV_INFO_X_WINDOW_SEGMENTS] = xwin ? xwin_segments : 0;
    {   // entry slab of the multi-vector kernel: shards with >= 3 entries per tile-row (strips then regularly hold more than the 16 entries that travel with the prologue)
        int used = 1;
        for (const STask &k : tasks) used = std::max(used, k.nrows);
        const int slab_env = env_int("TILESPMV_MV_SLAB", -1);   // (experiment knob: 0 off, 1 on wherever entries exist)
        plan->mv_slab_rows = (slab_env == 0 || NC == 0) ? 0 : (slab_env > 0 || NC >= 3LL * ntr) ? used : 0;
    }
    plan->mv_by_columns = entry_dominated && target >= 800;   // (small strips hold few entries each: scircuit-like 18 / 22 / 32 us native against 22 / 41 / 78 us)
    n_tasks = (long long)tasks.size();
    model_bytes = NUP * ((S.cb_bits > 0 ? 4 : 12) + 16LL * sv) + (entry_mode == 0 ? NC * (sv + 5LL) : n_rec * (long long)sizeof(ERec) + n_chunk * 4 + n_groups * 16) + NH * 8 + NHV * sv + NHI + n_tasks * (long long)sizeof(STask) +
                  (long long)htasks.size() * ((long long)sizeof(Task) + 32LL * sv) +  // whole-tile passes re-read and re-write their rows of y
                  ND * (4 + 256LL * sv) + (long long)drows.size() * (16 + 32LL * sv);
    // The once-read streams (values, entry records) are loaded nontemporally when the launch moves clearly more than the Infinity Cache holds: they then do
    // not displace x in the L2s / the Infinity Cache — config 4 0.182 -> 0.164-0.166 ms, 7-pt 256^3 0.231 -> 0.211, KKT fp32 0.253 -> 0.236-0.243, power-law
    // 8 M 0.103 -> 0.095 — while a plan that (nearly) fits keeps the default policy, because its streams come back from the Infinity Cache on the next SpMV:
    // nontemporal loses 2 % at 340 MB (5-pt 2400^2), 15 % at 180-300 MB (power-law 3-5 M rows), 6-8 % on webbase-1M; it wins from 500 MB up (5-pt 2896^2 +4 %,
    // power-law 8 M +8 %, 5-pt 3400^2 +10 %).  Descriptors, tasks and per-strip entry lists stay on the default policy (nontemporal: config 4 0.164 -> 0.170-0.173).
    // profiles/r03_nontemporal_streams.txt.  Entry mode 1 = small grids; x-window plans are an opt-in experiment.
    {
        const long long launch_b = model_bytes + ((long long)colA + 16LL * ntr) * sv;
        S.nt_stream = (entry_mode != 1 && !xwin && (K.nt_stream >= 0 ? K.nt_stream != 0 : launch_b > NT_STREAM_MIN_BYTES)) ? 1 : 0;
    }
    plan->info[TILESPMV_INFO_NT_STREAM] = S.nt_stream;
    return rc;
}

extern "C" {

int tilespmv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tilespmv_sizeof_value(void) { return (int)sizeof(val_t); }
const char *tilespmv_version(void) { return "tilespmv-mi355x 0.1 (gfx950)"; }

void tilespmv_partition_tilerows(const Tile_matrix *T, int nparts, int *bounds)
{
    // contiguous tile-row blocks balanced by stored payload (blknnz prefix, reference src/format.h:12)
    const long long total = T->blknnz[T->tilenum];
    bounds[0] = 0;
    int bi = 0;
    for (int p = 1; p < nparts; p++) {
        const long long want = total * p / nparts;
        while (bi < T->tilem && T->blknnz[T->tile_ptr[bi]] < want) bi++;
        bounds[p] = std::max(bi, bounds[p - 1]);
    }
    bounds[nparts] = T->tilem;
}

void tilespmv_plan_destroy(tilespmv_plan *plan)
{
    if (!plan) return;
    for (void *p : plan->allocs) (void)hipFree(p);
    delete plan;
}

static int plan_create_one(tilespmv_plan **out, const Tile_matrix *T, int rowA, int colA, MAT_PTR_TYPE nnzA, const Knobs &K);

void tilespmv_plan_options_init(tilespmv_plan_options *o)
{
    memset(o, 0, sizeof(*o));
    o->size = (unsigned)sizeof(*o);
    int *knob = &o->entry_mode;   // every field from entry_mode on is a knob
    const int n = (int)((sizeof(*o) - offsetof(tilespmv_plan_options, entry_mode)) / sizeof(int));
    for (int i = 0; i < n; i++) knob[i] = TILESPMV_KNOB_DEFAULT;
}

// "name:offset,name:offset,..." of every field of tilespmv_plan_options as THIS build lays it out: what a binding that mirrors the
// struct by hand (tilespmv_amd/_lib.py) is checked against (tests/test_plan_layout.py) — a permuted mirror used to go unnoticed.
const char *tilespmv_plan_options_layout(void)
{
    static const std::string s = [] {
        std::string o;
#define TSPMV_F(f) o += std::string(o.empty() ? "" : ",") + #f + ":" + std::to_string(offsetof(tilespmv_plan_options, f));
        TSPMV_F(size) TSPMV_F(coo_mode) TSPMV_F(dense_mode) TSPMV_F(kernel) TSPMV_F(tilerow_begin) TSPMV_F(tilerow_end) TSPMV_F(autotune)
        TSPMV_F(entry_mode) TSPMV_F(entry_ordered) TSPMV_F(strip_cost) TSPMV_F(split_above) TSPMV_F(split_cap) TSPMV_F(xcd_remap) TSPMV_F(xcd_chunk)
        TSPMV_F(csr_split) TSPMV_F(fix_inline) TSPMV_F(coo_cost) TSPMV_F(coo_heavy_min) TSPMV_F(coo_piece) TSPMV_F(strip_even) TSPMV_F(wg_strips)
        TSPMV_F(x_window) TSPMV_F(x_stride1) TSPMV_F(x_stride2) TSPMV_F(mv_native) TSPMV_F(mv_xcd_chunk) TSPMV_F(lds_pad) TSPMV_F(y_store)
        TSPMV_F(desc_dict) TSPMV_F(nt_stream) TSPMV_F(reserved)
#undef TSPMV_F
        return o;
    }();
    return s.c_str();
}

// Measured selection (execution side): with opts->autotune (or TILESPMV_AUTOTUNE=1) the choices that AUTO
// otherwise makes from byte models — COO tiles in-tile vs CSR fallback, dense tiles on the matrix cores vs as streamed
// units, entry mode, strip size, workgroup -> XCD map — are decided by timing each candidate plan on this device.  A
// candidate is a copy of the caller's Knobs with some fields replaced: nothing travels through the environment.
int tilespmv_plan_create(tilespmv_plan **out, const Tile_matrix *T, int rowA, int colA, MAT_PTR_TYPE nnzA,
                         const tilespmv_plan_options *opts)
{
    const Knobs K0 = resolve_knobs(opts);
    if (!K0.autotune) return plan_create_one(out, T, rowA, colA, nnzA, K0);
    *out = nullptr;
    const int tilem = T->tilem;
    const int tr0 = std::max(0, K0.tilerow_begin), tr1 = (K0.tilerow_end <= 0 || K0.tilerow_end > tilem) ? tilem : K0.tilerow_end;
    bool has_dense = false;
    for (int t = T->tile_ptr[tr0]; t < T->tile_ptr[tr1] && !has_dense; t++) has_dense = T->Format[t] == TILESPMV_FMT_DNS;
    const bool has_extracted = T->new_coocount[T->tile_ptr[tr1]] > T->new_coocount[T->tile_ptr[tr0]];
    val_t *dx = nullptr, *dy = nullptr;
    if (hipMalloc((void **)&dx, ((size_t)colA + 16) * sizeof(val_t)) != hipSuccess) return -3;
    if (hipMalloc((void **)&dy, ((size_t)rowA + 16) * sizeof(val_t)) != hipSuccess) { (void)hipFree(dx); return -3; }
    {   // x = 1 (not zero: candidates are timed on data that makes every product count)
        std::vector<val_t> ones((size_t)colA + 16, (val_t)1);
        (void)hipMemcpy(dx, ones.data(), ones.size() * sizeof(val_t), hipMemcpyHostToDevice);
    }
    tilespmv_plan *best = nullptr;
    double best_ms = 0;
    std::string log = "{\"rows\": " + std::to_string(rowA) + ", \"cols\": " + std::to_string(colA) + ", \"nnz\": " + std::to_string((long long)nnzA) +
                      ", \"value_bytes\": " + std::to_string(sizeof(val_t)) + ", \"candidates\": [";
    bool first = true;
    auto try_one = [&](const Knobs &cand, const char *label) {
        tilespmv_plan *p = nullptr;
        if (plan_create_one(&p, T, rowA, colA, nnzA, cand) != 0 || !p) return;
        const double ms = tilespmv_plan_time(p, dx, dy, nullptr, 3, 12);
        char buf[512];
        snprintf(buf, sizeof(buf), "%s{\"label\": \"%s\", \"coo_mode\": %d, \"dense_mode\": %d, \"entry_mode\": %d, \"ordered\": %lld, \"strip_cost\": %lld, \"tasks\": %lld, \"ms\": %.5f}",
                 first ? "" : ", ", label, p->coo_mode, p->dense_mode, p->entry_mode, p->info[TILESPMV_INFO_ENTRY_ORDERED], p->info[TILESPMV_INFO_STRIP_COST],
                 p->info[TILESPMV_INFO_NUM_TASKS], ms);
        log += buf; first = false;
        if (ms > 0 && (!best || ms < best_ms * 0.985)) { tilespmv_plan_destroy(best); best = p; best_ms = ms; }  // a later candidate must be clearly better
        else tilespmv_plan_destroy(p);
    };
    const int coo_cands[2] = {TILESPMV_COO_IN_TILE, TILESPMV_COO_FALLBACK}, dns_cands[2] = {TILESPMV_DENSE_MFMA, TILESPMV_DENSE_VALU};
    try_one(K0, "default");   // what AUTO picks from its byte models: stays unless something is clearly faster
    const tilespmv_plan *dflt = best;
    const int d_coo = dflt ? dflt->coo_mode : 0, d_dns = dflt ? dflt->dense_mode : 0, d_entry = dflt ? dflt->entry_mode : 0;
    const long long d_cost = dflt ? dflt->info[TILESPMV_INFO_STRIP_COST] : 400, d_ord = dflt ? dflt->info[TILESPMV_INFO_ENTRY_ORDERED] : 1;  // (copies: `best` may be replaced)
    const bool coo_var = K0.coo_mode == TILESPMV_COO_AUTO && has_extracted, dns_var = K0.dense_mode == TILESPMV_DENSE_AUTO && has_dense;
    for (int ci = 0; ci < (coo_var ? 2 : 1); ci++)
        for (int di = 0; di < (dns_var ? 2 : 1); di++) {
            Knobs cand = K0;
            if (coo_var) cand.coo_mode = coo_cands[ci];
            if (dns_var) cand.dense_mode = dns_cands[di];
            if ((!coo_var || cand.coo_mode == d_coo) && (!dns_var || cand.dense_mode == d_dns)) continue;   // that is the default, already timed
            try_one(cand, "coo/dense mode");
        }
    // how the entry lists run and how large the strips are (generation 2, in-tile entries): the other entry modes at the
    // default strip size, then the winning mode at half and twice the size.  Unordered workgroup adds are only a candidate
    // when the caller has not asked for reproducible sums (entry_ordered = 1).
    if (best && best->kernel == TILESPMV_KERNEL_STREAM && best->coo_mode == TILESPMV_COO_IN_TILE && !K0.entry_from_caller && !K0.strip_from_caller) {
        Knobs cand = K0;
        cand.coo_mode = best->coo_mode; cand.dense_mode = best->dense_mode;
        const bool may_unorder = K0.entry_ordered != 1;
        for (int em = 0; em <= 2; em++) {
            if (em == d_entry && !(em == 2)) continue;
            Knobs c2 = cand; c2.entry_mode = em;
            if (em == 2) {
                if (d_entry != 2 || d_ord == 0) { c2.entry_ordered = 1; try_one(c2, "entry mode 2, ordered"); }
                if (may_unorder && (d_entry != 2 || d_ord == 1)) { c2.entry_ordered = 0; try_one(c2, "entry mode 2, unordered"); }
            } else try_one(c2, em == 0 ? "entry mode 0" : "entry mode 1");
        }
        const int w_entry = best->entry_mode; const long long w_ord = best->info[TILESPMV_INFO_ENTRY_ORDERED], w_cost = best->info[TILESPMV_INFO_STRIP_COST];
        for (long long c : {w_cost / 2, w_cost * 2}) {
            if (c < 100 || c > 3200 || c == d_cost) continue;
            Knobs c2 = cand; c2.entry_mode = w_entry; c2.strip_cost = (int)c;
            if (w_entry == 2) c2.entry_ordered = w_ord ? 1 : 0;
            try_one(c2, "strip size");
        }
    }
    log += "], \"xcd_maps\": [";
    // the workgroup -> XCD mapping is a launch parameter: time the alternatives on the winning plan
    if (best && best->kernel == TILESPMV_KERNEL_STREAM && !K0.xcd_from_caller) {
        const int maps[3][2] = {{best->xcd_remap, best->xcd_chunk}, {0, best->xcd_chunk}, {2, 8}};
        int pick = 0;
        double pick_ms = 0;
        for (int k = 0; k < 3; k++) {
            best->xcd_remap = maps[k][0]; best->xcd_chunk = maps[k][1];
            const double ms = tilespmv_plan_time(best, dx, dy, nullptr, 3, 20);
            char buf[128];
            snprintf(buf, sizeof(buf), "%s{\"remap\": %d, \"chunk\": %d, \"ms\": %.5f}", k ? ", " : "", maps[k][0], maps[k][1], ms);
            log += buf;
            if (ms > 0 && (k == 0 || ms < pick_ms * 0.985)) { pick = k; pick_ms = ms; }  // leave the default unless clearly better
        }
        best->xcd_remap = maps[pick][0]; best->xcd_chunk = maps[pick][1];
        best_ms = std::min(best_ms, pick_ms);
    }
    // cache policy of the once-read streams: the size rule (nontemporal above 400 MB per launch) switches somewhere between 340 and 500 MB; a measured selection
    // just times both (the kernels differ by a template flag only: nothing is rebuilt).  Entry mode 1 and x-window plans have no nontemporal form.
    log += "], \"stream_policy\": [";
    if (best && best->kernel == TILESPMV_KERNEL_STREAM && K0.nt_stream < 0 && best->entry_mode != 1 && best->xwin_lds_bytes == 0) {
        const int rule = best->st.nt_stream;
        double ms2[2] = {0, 0};
        for (int k = 0; k < 2; k++) {
            best->st.nt_stream = k ? 1 - rule : rule;
            ms2[k] = tilespmv_plan_time(best, dx, dy, nullptr, 3, 20);
            char buf[96];
            snprintf(buf, sizeof(buf), "%s{\"nontemporal\": %d, \"ms\": %.5f}", k ? ", " : "", best->st.nt_stream, ms2[k]);
            log += buf;
        }
        const bool flip = ms2[1] > 0 && ms2[1] < ms2[0] * 0.985;   // leave the rule unless clearly better
        best->st.nt_stream = flip ? 1 - rule : rule;
        best->info[TILESPMV_INFO_NT_STREAM] = best->st.nt_stream;
        if (ms2[flip ? 1 : 0] > 0) best_ms = std::min(best_ms, ms2[flip ? 1 : 0]);
    }
    if (best) {
        char buf[512];
        snprintf(buf, sizeof(buf), "], \"choice\": {\"coo_mode\": %d, \"dense_mode\": %d, \"entry_mode\": %d, \"ordered\": %lld, \"strip_cost\": %lld, \"xcd_remap\": %d, \"xcd_chunk\": %d, \"nontemporal\": %d, \"ms\": %.5f}}",
                 best->coo_mode, best->dense_mode, best->entry_mode, best->info[TILESPMV_INFO_ENTRY_ORDERED], best->info[TILESPMV_INFO_STRIP_COST], best->xcd_remap, best->xcd_chunk, best->st.nt_stream, best_ms);
        log += buf;
        if (K0.autotune_log) {   // one JSON line per tuned plan
            if (FILE *f = fopen(K0.autotune_log, "a")) { fprintf(f, "%s\n", log.c_str()); fclose(f); }
        }
    }
    (void)hipFree(dx); (void)hipFree(dy);
    if (!best) return -4;
    *out = best;
    return 0;
}

static int plan_create_one(tilespmv_plan **out, const Tile_matrix *T, int rowA, int colA, MAT_PTR_TYPE nnzA, const Knobs &K)
{
    (void)nnzA;
    *out = nullptr;
    const double t_create0 = now_us();
    if (!K.dry && tilespmv_device_count() <= 0) {
        fprintf(stderr, "tilespmv: no HIP device visible — the GPU path has no CPU fallback\n");
        return -1;
    }
    const Knobs &o = K;
    const int tilem = T->tilem, tilen = T->tilen;
    const int tr0 = std::max(0, o.tilerow_begin), tr1 = (o.tilerow_end <= 0 || o.tilerow_end > tilem) ? tilem : o.tilerow_end;
    const int ntr = std::max(0, tr1 - tr0);
    const int sv = (int)sizeof(val_t);

    auto *plan = new tilespmv_plan();
    plan->dry = K.dry;
    if (const char *af = getenv("TILESPMV_ARENA_FLAGS")) plan->arena_flags = atoi(af);
    if (const char *as = getenv("TILESPMV_ARENA_SKEW")) plan->arena_skew = (size_t)std::max(0ll, atoll(as)) / 256 * 256;
    if (const char *ab = getenv("TILESPMV_ARENA_MB")) plan->arena_block = (size_t)std::max(0, atoi(ab)) << 20;   // (experiment knob; 0 = one hipMalloc per stream)
    if (!K.dry && hipGetDevice(&plan->device) != hipSuccess) { fprintf(stderr, "tilespmv: hipGetDevice failed\n"); delete plan; return -1; }

    // ---- how are COO tiles executed?  (bytes model)
    const int t_begin = T->tile_ptr[tr0], t_end = T->tile_ptr[tr1];
    const long long shard_rows = std::min<long long>((long long)tr1 * 16, rowA) - (long long)tr0 * 16;
    long long ncoo_tiles = 0, ncoo_vals = 0;
    for (int t = t_begin; t < t_end; t++)
        if (T->Format[t] == TILESPMV_FMT_COO) { ncoo_tiles++; ncoo_vals += T->blknnz[t + 1] - T->blknnz[t]; }
    const long long extracted = T->new_coocount[t_end] - T->new_coocount[t_begin];
    int coo_mode = o.coo_mode, dense_mode = o.dense_mode, kernel = o.kernel;
    if (kernel == TILESPMV_KERNEL_AUTO)  // the unit descriptor keeps the column block in 24 bits
        kernel = tilen <= (1 << UNIT_FLAG_SHIFT) ? TILESPMV_KERNEL_STREAM : TILESPMV_KERNEL_DIRECT;
    if (coo_mode == TILESPMV_COO_AUTO) {
        if (kernel == TILESPMV_KERNEL_STREAM) {
            // unit-stream kernel: the in-tile entry list is the extracted matrix folded into the fused launch (s_v + 5 bytes per
            // nonzero, no per-tile descriptor); the fallback moves the same bytes plus a second launch and its rows of y twice.
            // It never wins (1.1-2x behind on every measured matrix, 1.1-1.4x on uniform random ones).
            coo_mode = TILESPMV_COO_IN_TILE;
        } else {
            const long long in_tile = ncoo_tiles * 8 + ncoo_vals * (sv + 1);   // generation 1: 8-byte descriptor per COO tile
            const long long fallback = extracted * (sv + 5) + shard_rows * 2 * sv;
            coo_mode = (extracted > 0 && fallback * 10 < in_tile * 9) ? TILESPMV_COO_FALLBACK : TILESPMV_COO_IN_TILE;
        }
    }
    const bool coo_in_tile = coo_mode == TILESPMV_COO_IN_TILE;
    // Dense tiles: the matrix-core routine handles one tile per wavefront at a time; in the unit
    // kernel a dense tile is 16 streamed units instead, which measures faster on MI355X
    // (DESIGN.md §5), so AUTO keeps MFMA for the tile-at-a-time kernel only.
    if (dense_mode == TILESPMV_DENSE_AUTO) {
        if (kernel != TILESPMV_KERNEL_STREAM) dense_mode = TILESPMV_DENSE_MFMA;
        else {
            // generation 2: dense tiles run on the matrix cores in their own pass (k_dense_mfma, one wavefront per tile-row,
            // accumulator carried across the row's dense tiles) when they carry a real share of the payload AND a tile-row
            // holds several of them (band hbw 40, 5 per row: 0.263 ms vs 0.30-0.32 ms as streamed units; band hbw 12, one per
            // row: 0.066 ms vs 0.050 ms — a one-tile chain does not pay for the extra launch and its y update); a handful
            // of dense tiles is cheaper as 16 units each
            long long dense_vals = 0, dense_rows = 0;
            for (int bi = tr0; bi < tr1; bi++) {
                long long nd = 0;
                for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) if (T->Format[t] == TILESPMV_FMT_DNS) nd++;
                dense_vals += 256 * nd; dense_rows += nd > 0;
            }
            const long long all_vals = (long long)T->blknnz[t_end] - T->blknnz[t_begin];
            dense_mode = (dense_vals * 10 >= all_vals && dense_vals >= 256 * 1024 && dense_vals >= 256 * 5 * dense_rows / 2) ? TILESPMV_DENSE_MFMA : TILESPMV_DENSE_VALU;
        }
    }
    plan->coo_mode = coo_mode; plan->dense_mode = dense_mode;

    // ---- HYB tiles address hybIdx by a running byte offset (reference ptroffset2, src/tilespmv_cpu.h:195-196)
    std::vector<long long> hyb_off;
    if (T->hybsize > 0) {
        hyb_off.assign((size_t)T->tilenum, 0);
        long long at = 0;
        for (int bi = 0; bi < tilem; bi++) {
            const int rowlen = tile_rowlen(bi, tilem, rowA);
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++)
                if (T->Format[t] == TILESPMV_FMT_HYB) {
                    hyb_off[t] = at;
                    const int nell = T->tilewidth[t] * rowlen;
                    at += (nell + 1) / 2 + (T->blknnz[t + 1] - T->blknnz[t] - nell);
                }
        }
    }

    plan->kernel = kernel;
    plan->xcd_remap = K.xcd_remap;  // 0 = round-robin, 2 = windows of 8 x xcd_chunk workgroups
    plan->xcd_chunk = K.xcd_chunk;
    plan->mv_native = K.mv_native; plan->mv_xcd_chunk = K.mv_xcd_chunk;
    plan->lds_pad_bytes = K.lds_pad > 0 ? std::min(K.lds_pad, 40 * 1024) : 0;
    // Strip size: ~400 cost units (20 units) amortises the per-strip round trips; measured flat between 200
    // and 800 on large matrices and neutral on small (cache-resident) ones, where launch latency dominates.
    const int target_env = K.strip_cost, split_env = K.split_above;
    int target = target_env;    // generation 1 below; the unit-stream builder picks its own default (build_stream)
    if (target <= 0) target = 192;
    target = std::max(32, target);
    const int split_above = std::max(6 * target, split_env), piece = std::max(2 * target, split_above / 3);
    std::vector<FixRow> fix;
    int npartial = 0;
    long long n_tasks = 0, model_bytes = 0;
    int rc = 0;
    DevPlan &D = plan->dev;
    if (kernel == TILESPMV_KERNEL_STREAM) {
        rc = build_stream(plan, K, T, rowA, colA, tr0, tr1, coo_in_tile, dense_mode == TILESPMV_DENSE_MFMA, hyb_off, fix, npartial, n_tasks, model_bytes);
    } else {
    // ---- pass 1: stream sizes per tile-row
    std::vector<long long> row_tile((size_t)ntr + 1, 0), row_val((size_t)ntr + 1, 0), row_idx((size_t)ntr + 1, 0);
    parallel_chunks(ntr, 1024, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            const int bi = tr0 + (int)i, rowlen = tile_rowlen(bi, tilem, rowA);
            long long nt = 0, nv = 0, ni = 0;
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                Emit em = emit_of(T, t, rowlen, coo_in_tile);
                if (em.fmt == (int)DESC_FMT_NOP) continue;
                nt++; nv += em.nv; ni += em.ni;
            }
            if (nt == 0) nt = 1;  // placeholder so that the tile-row still writes its (zero) results
            row_tile[i + 1] = nt; row_val[i + 1] = nv; row_idx[i + 1] = ni;
        }
    });
    for (int i = 0; i < ntr; i++) { row_tile[i + 1] += row_tile[i]; row_val[i + 1] += row_val[i]; row_idx[i + 1] += row_idx[i]; }
    const long long n_desc = row_tile[ntr], n_val = row_val[ntr], n_idx = row_idx[ntr];
    if (n_desc > INT32_MAX) { fprintf(stderr, "tilespmv: shard has too many tiles\n"); delete plan; return -2; }

    // ---- pass 2: fill the streams
    std::vector<uint2> h_desc((size_t)n_desc);
    val_t *h_val = zalloc<val_t>((size_t)n_val);
    unsigned char *h_idx = zalloc<unsigned char>((size_t)n_idx + 16);
    std::vector<int> h_cost((size_t)n_desc);  // per emitted tile, for task cutting
    parallel_chunks(ntr, 256, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            const int bi = tr0 + (int)i, rowlen = tile_rowlen(bi, tilem, rowA);
            long long d = row_tile[i], vo = row_val[i], io = row_idx[i];
            const long long d0 = d;
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                Emit em = emit_of(T, t, rowlen, coo_in_tile);
                if (em.fmt == (int)DESC_FMT_NOP) continue;
                const int cb = T->tile_columnidx[t];
                repack_tile(T, t, em, rowlen, tile_collen(cb, tilen, colA), hyb_off.empty() ? 0 : hyb_off[t], h_val + vo, h_idx + io);
                h_desc[d] = make_uint2((unsigned)cb, (unsigned)em.fmt | ((unsigned)em.p1 << DESC_P1_SHIFT) | ((unsigned)em.p2 << DESC_P2_SHIFT));
                h_cost[d] = em.nv + 8;
                d++; vo += em.nv; io += em.ni;
            }
            if (d == d0) { h_desc[d] = make_uint2(0u, DESC_FMT_NOP); h_cost[d] = 4; d++; }
            h_desc[d - 1].y |= DESC_EOR;
        }
    });

    // ---- strips: consecutive whole tile-rows up to a cost target; very long tile-rows are cut
    // at tile boundaries into pieces whose partial sums are combined by k_fixup_split.
    std::vector<Task> tasks;
    {
        auto row_cost = [&](int i) { long long c = 0; for (long long d = row_tile[i]; d < row_tile[i + 1]; d++) c += h_cost[d]; return c; };
        int i = 0;
        while (i < ntr) {
            const long long c0 = row_cost(i);
            if (c0 > split_above) {
                FixRow f{tr0 + i, npartial, 0, 0};
                long long d = row_tile[i], vo = row_val[i], io = row_idx[i];
                while (d < row_tile[i + 1]) {
                    Task k{(int)d, (int)d, vo, io, tr0 + i, npartial++};
                    long long c = 0;
                    while (d < row_tile[i + 1] && (c == 0 || c + h_cost[d] <= piece)) {
                        const unsigned m = h_desc[d].y; int nv, ni;
                        tile_stream_sizes((int)(m & DESC_FMT_MASK), (int)((m >> DESC_P1_SHIFT) & 255u), (int)((m >> DESC_P2_SHIFT) & 255u), &nv, &ni);
                        c += h_cost[d]; vo += nv; io += ni; d++;
                    }
                    k.tile_end = (int)d;
                    tasks.push_back(k); f.count++;
                }
                fix.push_back(f);
                i++;
                continue;
            }
            Task k{(int)row_tile[i], 0, row_val[i], row_idx[i], tr0 + i, -1};
            long long c = 0;
            int j = i;
            while (j < ntr && (j == i || c + row_cost(j) <= target) && row_cost(j) <= split_above) { c += row_cost(j); j++; }
            k.tile_end = (int)row_tile[j];
            tasks.push_back(k);
            i = j;
        }
    }

    rc |= plan->upload(h_desc.data(), (size_t)n_desc, &D.desc);
    rc |= plan->upload(h_val, (size_t)n_val, &D.val);
    rc |= plan->upload(h_idx, (size_t)n_idx, &D.idx);
    rc |= plan->upload(tasks.data(), tasks.size(), &D.task);
    free(h_val); free(h_idx);
    D.ntasks = (int)tasks.size();
    n_tasks = (long long)tasks.size();
    model_bytes = n_desc * 8 + n_val * sv + n_idx + n_tasks * (long long)sizeof(Task);
    }
    // ---- very-sparse fallback matrix: the shard's rows of deferredcoo_*, cut into nnz-balanced row blocks (<= FB_CAP nonzeros,
    // <= FB_ROWS rows, one workgroup each; a single longer row becomes several one-row pieces that add atomically), the nonzeros
    // of a block ordered by column with their row-in-block split over the column word's top bits and the row byte (hip_plan.h)
    std::vector<int4> f_blk;
    long long f_nnz = 0;
    const int row0 = tr0 * 16, rows = (int)shard_rows;
    std::vector<ERec> f_rec;
    std::vector<unsigned> f_base;
    if (!coo_in_tile && extracted > 0) {
        const int *P0 = T->deferredcoo_ptr + row0;
        const int base = P0[0];
        f_nnz = P0[rows] - base;
        // block size: FB_CAP nonzeros when there is enough work for ~3 workgroups per CU, smaller blocks (down to one trip) otherwise
        const int cap = (int)std::max<long long>(1536, std::min<long long>(FB_CAP, f_nnz / (3 * 256)));
        int r = 0;
        while (r < rows) {   // .z / .w: source range in the extracted matrix for now, record range after packing
            const int nr = P0[r + 1] - P0[r];
            if (nr == 0) { r++; continue; }   // runs of empty rows are not covered at all
            if (nr > FB_CAP) {
                for (int sft = P0[r]; sft < P0[r + 1]; sft += FB_CAP) f_blk.push_back(make_int4(r, -1, sft - base, std::min(sft + FB_CAP, P0[r + 1]) - base));
                r++;
                continue;
            }
            int e = r, cnt = 0;
            while (e < rows && e - r < FB_ROWS && (P0[e + 1] - P0[e]) <= FB_CAP && (e == r || cnt + (P0[e + 1] - P0[e]) <= cap)) { cnt += P0[e + 1] - P0[e]; e++; }
            while (e > r + 1 && P0[e] == P0[e - 1]) e--;  // drop trailing empty rows
            f_blk.push_back(make_int4(r, e - r, P0[r] - base, P0[e] - base));
            r = e;
        }
        std::vector<std::vector<ERec>> blk_rec(f_blk.size());
        std::vector<std::vector<unsigned>> blk_base(f_blk.size());
        std::atomic<int> bad(0);
        parallel_chunks((int64_t)f_blk.size(), 16, [&](int64_t b0, int64_t b1, int) {
            std::vector<std::pair<unsigned long long, int>> key;
            std::vector<PEnt> ents;
            for (int64_t b = b0; b < b1; b++) {
                const int4 k = f_blk[(size_t)b];
                key.clear();
                if (k.y < 0) {
                    for (int q = k.z; q < k.w; q++) key.push_back({((unsigned long long)(unsigned)T->deferredcoo_colidx[base + q] << 32) | (unsigned)(q - k.z), 0});
                } else {
                    for (int rr = k.x; rr < k.x + k.y; rr++)
                        for (int q = P0[rr] - base; q < P0[rr + 1] - base; q++)
                            key.push_back({((unsigned long long)(unsigned)T->deferredcoo_colidx[base + q] << 32) | (unsigned)(q - k.z), rr - k.x});
                }
                std::sort(key.begin(), key.end());   // by column; ties keep the extracted matrix's order
                ents.resize(key.size());
                for (size_t i = 0; i < key.size(); i++) {
                    const int q = k.z + (int)(key[i].first & 0xFFFFFFFFull);
                    ents[i] = PEnt{(unsigned)T->deferredcoo_colidx[base + q], (unsigned)key[i].second, T->deferredcoo_val[base + q]};
                }
                if (!pack_list(ents, FB_DEST_BITS, blk_rec[(size_t)b], blk_base[(size_t)b], K.dry)) bad++;
            }
        });
        if (bad.load()) { fprintf(stderr, "tilespmv: internal error: %d packed fallback lists do not decode to their entries\n", bad.load()); rc = -6; }
        long long at = 0;   // every block's list starts on a chunk boundary: chunk of record i = i >> 6
        for (size_t b = 0; b < f_blk.size(); b++) {
            f_blk[b].z = (int)at; f_blk[b].w = (int)(at + (long long)blk_rec[b].size());
            at = (at + (long long)blk_rec[b].size() + ECHUNK - 1) / ECHUNK * ECHUNK;
            if (at > INT32_MAX - ECHUNK) { fprintf(stderr, "tilespmv: shard too large for 32-bit entry ids\n"); rc = -2; break; }
        }
        if (rc == 0) {
            f_rec.assign((size_t)at, make_erec((val_t)0, 0u));
            f_base.assign((size_t)(at / ECHUNK), 0u);
            parallel_chunks((int64_t)f_blk.size(), 64, [&](int64_t b0, int64_t b1, int) {
                for (int64_t b = b0; b < b1; b++) {
                    if (!blk_rec[(size_t)b].empty()) memcpy(&f_rec[(size_t)f_blk[(size_t)b].z], blk_rec[(size_t)b].data(), blk_rec[(size_t)b].size() * sizeof(ERec));
                    if (!blk_base[(size_t)b].empty()) memcpy(&f_base[(size_t)(f_blk[(size_t)b].z / ECHUNK)], blk_base[(size_t)b].data(), blk_base[(size_t)b].size() * sizeof(unsigned));
                }
            });
        }
    }

    // ---- upload the rest
    rc |= plan->upload(fix.data(), fix.size(), &D.fix);
    if (npartial > 0) {
        void *p = nullptr;
        if (K.dry) { }
        else if (hipMalloc(&p, (size_t)npartial * 16 * sizeof(val_t) * TILESPMV_MAX_NVEC) != hipSuccess) rc = -3;  // slots are nvec wide in tilespmv_plan_spmm
        else { plan->allocs.push_back(p); D.partial = (val_t *)p; }
    }
    if (!f_blk.empty()) {
        rc |= plan->upload(f_rec.data(), f_rec.size(), &D.f_rec);
        rc |= plan->upload(f_base.data(), f_base.size(), &D.f_base);
        rc |= plan->upload(f_blk.data(), f_blk.size(), &D.f_blk);
        D.f_nblk = (int)f_blk.size();
        // taking turns costs latency on small grids and nothing on large ones (as in the unit kernel's workgroup entry mode)
        const int ordered_env = K.entry_ordered;
        D.f_ordered = ordered_env >= 0 ? ordered_env != 0 : f_blk.size() >= 2048;
    }
    if (rc) { tilespmv_plan_destroy(plan); return rc; }
    D.nfix = (int)fix.size();
    D.rowA = std::min<long long>(rowA, (long long)tr1 * 16); D.colA = colA;
    D.f_row0 = row0; D.f_rows = rows;

    long long *I = plan->info;
    I[TILESPMV_INFO_NNZ] = T->tile_nnz[t_end] - T->tile_nnz[t_begin];
    I[TILESPMV_INFO_ROWS] = rows;
    I[TILESPMV_INFO_TILES] = t_end - t_begin;
    I[TILESPMV_INFO_COO_MODE] = coo_mode; I[TILESPMV_INFO_DENSE_MODE] = dense_mode; I[TILESPMV_INFO_KERNEL] = plan->kernel;
    I[TILESPMV_INFO_NUM_TASKS] = n_tasks; I[TILESPMV_INFO_NUM_SPLIT_ROWS] = (long long)fix.size();
    I[TILESPMV_INFO_FALLBACK_NNZ] = f_nnz;
    if (plan->kernel != TILESPMV_KERNEL_STREAM) { I[TILESPMV_INFO_ENTRY_ORDERED] = 1; I[TILESPMV_INFO_STRIP_COST] = target; }
    I[TILESPMV_INFO_BUILD_US] = (long long)(now_us() - t_create0) - I[TILESPMV_INFO_UPLOAD_US];
    // bytes one SpMV has to move at least: the three streams + tasks + x once + y once (+ fallback)
    I[TILESPMV_INFO_STREAM_BYTES] = model_bytes + (long long)colA * sv + (long long)rows * sv +
                                    (f_nnz ? (long long)f_rec.size() * (long long)sizeof(ERec) + (long long)f_base.size() * 4 + 2LL * sv * rows + (long long)f_blk.size() * 16 : 0);   // the fallback re-reads and re-writes its rows of y
    *out = plan;
    return 0;
}

int tilespmv_plan_layout_digest(const Tile_matrix *T, int rowA, int colA, MAT_PTR_TYPE nnzA, const tilespmv_plan_options *opts,
                                unsigned long long *digest, long long *info)
{
    Knobs K = resolve_knobs(opts);
    K.dry = true; K.autotune = 0;
    tilespmv_plan *p = nullptr;
    const int rc = plan_create_one(&p, T, rowA, colA, nnzA, K);
    if (rc != 0 || !p) return rc ? rc : -4;
    if (digest) *digest = p->digest;
    if (info) memcpy(info, p->info, sizeof(p->info));
    tilespmv_plan_destroy(p);
    return 0;
}

int tilespmv_plan_spmv(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = plan->dense_mode == TILESPMV_DENSE_MFMA;
    hipError_t e = plan->kernel == TILESPMV_KERNEL_STREAM ? launch_tiles_stream(plan->dev, plan->st, plan->dn, mfma, plan->entry_mode, plan->wg_strips, plan->xwin_lds_bytes, plan->lds_pad_bytes, plan->xcd_remap, plan->xcd_chunk, d_x, d_y, st)
                                                          : launch_tiles_direct(plan->dev, mfma, false, true, d_x, d_y, st);
    if (e != hipSuccess) return (int)e;
    return (int)launch_fallback(plan->dev, d_x, d_y, st);
}

int tilespmv_plan_reserve_spmm(tilespmv_plan *plan, int nvec)
{
    if (nvec < 1 || nvec > TILESPMV_MAX_NVEC) return (int)hipErrorInvalidValue;
    if (plan->mv_nvec >= nvec) return 0;
    const long long rows = plan->dev.f_rows, ldx = ((long long)plan->dev.colA + 16 + 15) / 16 * 16, ldy = (rows + 16 + 15) / 16 * 16;
    void *px = nullptr, *py = nullptr;
    if (hipMalloc(&px, (size_t)ldx * nvec * sizeof(val_t)) != hipSuccess) return (int)hipErrorOutOfMemory;
    if (hipMalloc(&py, (size_t)ldy * nvec * sizeof(val_t)) != hipSuccess) { (void)hipFree(px); return (int)hipErrorOutOfMemory; }
    for (void *old : {(void *)plan->mv_x, (void *)plan->mv_y})   // a narrower pair from an earlier call goes back
        if (old) { (void)hipFree(old); plan->allocs.erase(std::find(plan->allocs.begin(), plan->allocs.end(), old)); }
    plan->allocs.push_back(px); plan->allocs.push_back(py);
    plan->mv_x = (val_t *)px; plan->mv_y = (val_t *)py; plan->mv_nvec = nvec;
    return 0;
}

int tilespmv_plan_spmm(tilespmv_plan *plan, const MAT_VAL_TYPE *d_X, MAT_VAL_TYPE *d_Y, int nvec, void *stream)
{
    if (nvec == 1) return tilespmv_plan_spmv(plan, d_X, d_Y, stream);
    if (nvec != 2 && nvec != 4 && nvec != 8) return (int)hipErrorInvalidValue;
    if (((uintptr_t)d_X | (uintptr_t)d_Y) & 15u) return (int)hipErrorInvalidValue;  // rows of X / Y travel as 16-B vectors
    // native multi-vector kernels: unit-stream plans whose COO entries run in-tile and whose CSR tiles were split into units
    // (the defaults).  Generation-1 plans, whole-tile passes and the CSR fallback go one right-hand side at a time.
    const int mv_native = plan->mv_native;   // 1 / 0: force the multi-vector kernel / the one-at-a-time path on entry-dominated plans
    const bool has_native = plan->kernel == TILESPMV_KERNEL_STREAM && plan->dev.ntasks == 0 && plan->dev.f_nblk == 0;
    // entry-dominated plans (round 3, final): the multi-vector kernel scatters a strip's entries up front (entry slab, mv_slab_rows), which beats going one right-hand side
    // at a time at every nvec (webbase stand-in 37 / 58 / 93 us against 42 / 85 / 223) and beats the separate entry pass over the merged lists (k_entries_mv; workgroup entry mode,
    // 16 strips, no x windows) from nvec 4 on (power-law 8 M: 0.254 / 0.375 / 0.678 ms against 0.198 / 0.469 / 1.364 with the pass): the pass stays for nvec 2.
    // mv_native: -1 by rule, 0 one right-hand side at a time, 1 the multi-vector kernel alone, 2 the multi-vector kernel + entry pass
    const bool can_pass = has_native && plan->entry_mode == 2 && plan->wg_strips == 16 && plan->xwin_lds_bytes == 0;
    const bool entries_pass = can_pass && (mv_native == 2 || (mv_native < 0 && plan->mv_by_columns && nvec < 4));
    const bool one_at_a_time = !has_native || mv_native == 0 || (mv_native < 0 && plan->mv_by_columns && !entries_pass && plan->mv_slab_rows == 0 && nvec < 8);
    if (one_at_a_time) {
        hipStream_t st = (hipStream_t)stream;
        // leading dimensions of the column copies: multiples of 16 elements, so that every column of X / Y starts 64- / 128-byte
        // aligned whatever rowA is (the SpMV kernels store y with 16-byte lane stores)
        const long long row0 = plan->dev.f_row0, rows = plan->dev.f_rows, ldx = ((long long)plan->dev.colA + 16 + 15) / 16 * 16, ldy = (rows + 16 + 15) / 16 * 16;
        if (plan->mv_nvec < nvec) {   // not reserved (tilespmv_plan_reserve_spmm): allocate now — synchronises, and fails under stream capture
            const int rc = tilespmv_plan_reserve_spmm(plan, nvec);
            if (rc) return rc;
        }
        hipError_t e = launch_rows_to_columns(d_X, nvec, plan->dev.colA, ldx, plan->mv_x, st);
        if (e != hipSuccess) return (int)e;
        for (int j = 0; j < nvec; j++) {
            const int rc = tilespmv_plan_spmv(plan, plan->mv_x + j * ldx, plan->mv_y + j * ldy - row0, stream);  // the plan writes rows row0 .. row0 + rows of what it is handed
            if (rc) return rc;
        }
        e = launch_columns_to_rows(plan->mv_y, nvec, row0, rows, ldy, d_Y, st);
        return (int)e;
    }
    const int mv_chunk = plan->mv_xcd_chunk;
    return (int)launch_tiles_stream_mv(plan->dev, plan->st, plan->dn, nvec, mv_chunk >= 0 ? mv_chunk : (plan->xcd_remap >= 2 ? plan->xcd_chunk : 0), entries_pass, plan->mv_slab_rows, d_X, d_Y, (hipStream_t)stream);
}

double tilespmv_plan_time_spmm(tilespmv_plan *plan, const MAT_VAL_TYPE *d_X, MAT_VAL_TYPE *d_Y, int nvec, void *stream, int warmup, int reps)
{
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess) return -1.0;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1.0; }
    bool ok = true;
    for (int i = 0; ok && i < warmup; i++) ok = tilespmv_plan_spmm(plan, d_X, d_Y, nvec, stream) == 0;
    if (ok) ok = hipEventRecord(a, st) == hipSuccess;
    for (int i = 0; ok && i < reps; i++) ok = tilespmv_plan_spmm(plan, d_X, d_Y, nvec, stream) == 0;
    if (ok) ok = hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess;
    float ms = 0.f;
    if (ok) ok = hipEventElapsedTime(&ms, a, b) == hipSuccess;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (!ok) return -1.0;
    return reps > 0 ? (double)ms / reps : 0.0;
}

int tilespmv_plan_spmv_n(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y, void *stream, int count)
{
    for (int i = 0; i < count; i++) {
        const int rc = tilespmv_plan_spmv(plan, d_x, d_y, stream);
        if (rc) return rc;
    }
    return 0;
}

#ifdef TILESPMV_STAMPS
// diagnostic build only: copies the per-wavefront clock stamps of the last k_units launch (8 per wavefront) to the host
long long tilespmv_plan_stamps(const tilespmv_plan *plan, unsigned long long *out, long long max_words)
{
    const long long n = ((long long)plan->st.ntasks + 15) / 16 * 4 * 8;
    if (!plan->st.stamps || n > max_words) return -n;
    if (hipMemcpy(out, plan->st.stamps, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}
#endif

void tilespmv_plan_info(const tilespmv_plan *plan, long long *out)
{
    memcpy(out, plan->info, sizeof(plan->info));
}

double tilespmv_plan_time(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y, void *stream, int warmup, int reps)
{
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess) return -1.0;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1.0; }
    bool ok = true;
    for (int i = 0; ok && i < warmup; i++) ok = tilespmv_plan_spmv(plan, d_x, d_y, stream) == 0;
    if (ok) ok = hipEventRecord(a, st) == hipSuccess;
    for (int i = 0; ok && i < reps; i++) ok = tilespmv_plan_spmv(plan, d_x, d_y, stream) == 0;
    if (ok) ok = hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess;
    float ms = 0.f;
    if (ok) ok = hipEventElapsedTime(&ms, a, b) == hipSuccess;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (!ok) return -1.0;
    return reps > 0 ? (double)ms / reps : 0.0;
}

double tilespmv_plan_time_reference_style(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y, void *stream, int reps)
{
    // the reference's protocol (src/tilespmv_cuda.h:1112-1137): gettimeofday around ONE launch + synchronize, summed over the repeats
    hipStream_t st = (hipStream_t)stream;
    if (hipStreamSynchronize(st) != hipSuccess) return -1.0;
    double wall_ms = 0;
    for (int i = 0; i < reps; i++) {
        timeval t1, t2;
        gettimeofday(&t1, NULL);
        if (tilespmv_plan_spmv(plan, d_x, d_y, stream) != 0) return -1.0;
        if (hipStreamSynchronize(st) != hipSuccess) return -1.0;
        gettimeofday(&t2, NULL);
        wall_ms += (t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0;
    }
    return reps > 0 ? wall_ms / reps : 0.0;
}

void call_tilespmv_hip(char *filename, Tile_matrix *matrix, int *ptroffset1, int *ptroffset2, int rowblkblock,
                       unsigned int *blkcoostylerowidx, int *blkcoostylerowidx_colstart, int *blkcoostylerowidx_colstop,
                       int rowA, int colA, MAT_PTR_TYPE nnzA, MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA,
                       MAT_VAL_TYPE *csrValA, MAT_VAL_TYPE alpha, MAT_VAL_TYPE *x, MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden)
{
    // The reference's schedule arrays are pure functions of the Tile_matrix (SURVEY.md Appendix A
    // invariants); the plan derives its own strip schedule, so they are accepted and not used.
    (void)ptroffset1; (void)ptroffset2; (void)rowblkblock; (void)blkcoostylerowidx; (void)blkcoostylerowidx_colstart;
    (void)blkcoostylerowidx_colstop; (void)csrRowPtrA; (void)csrColIdxA; (void)csrValA; (void)alpha; (void)y_golden;
    auto die = [](const char *what, int code) { fprintf(stderr, "call_tilespmv_hip: %s failed (%d)\n", what, code); exit(3); };
    tilespmv_plan *plan = nullptr;
    int rc = tilespmv_plan_create(&plan, matrix, rowA, colA, nnzA, nullptr);
    if (rc) die("tilespmv_plan_create", rc);
    val_t *d_x = nullptr, *d_y = nullptr;
    if ((rc = hipMalloc((void **)&d_x, ((size_t)colA + 16) * sizeof(val_t)))) die("hipMalloc x", rc);
    if ((rc = hipMalloc((void **)&d_y, ((size_t)rowA + 16) * sizeof(val_t)))) die("hipMalloc y", rc);
    if ((rc = hipMemcpy(d_x, x, (size_t)colA * sizeof(val_t), hipMemcpyHostToDevice))) die("hipMemcpy x", rc);

    const int warm = env_int("TILESPMV_WARMUP", 200), reps = std::max(1, env_int("TILESPMV_BENCH_REPEAT", 1000));
    for (int i = 0; i < warm; i++) if ((rc = tilespmv_plan_spmv(plan, d_x, d_y, nullptr))) die("warm-up launch", rc);
    if ((rc = hipDeviceSynchronize())) die("hipDeviceSynchronize", rc);

    // reference-style number: wall clock around launch + sync, one SpMV at a time (src/tilespmv_cuda.h:1112-1137)
    const double wall_ms = tilespmv_plan_time_reference_style(plan, d_x, d_y, nullptr, reps);
    if (wall_ms < 0) die("timed launch", 1);
    const double gflops = 2 * (double)nnzA * 1.0e-6 / wall_ms;
    printf("  CUDA SpMV runtime %4.2f ms, %4.2f GFlops\n\n", wall_ms, gflops);

    // added line: device time of back-to-back launches (hipEvents) and the memory-roofline view
    const double ev_ms = tilespmv_plan_time(plan, d_x, d_y, nullptr, 0, reps);
    const double b_alg = (double)nnzA * (sizeof(val_t) + 4) + 4.0 * (rowA + 1) + (double)sizeof(val_t) * ((double)colA + rowA);
    printf("  HIP SpMV device time %.4f ms, %.2f GFlops, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n\n", ev_ms,
           2 * (double)nnzA * 1.0e-6 / ev_ms, b_alg * 1e-6 / ev_ms, b_alg * 1e-6 / ev_ms / 80.0);

    FILE *fout = fopen("results.csv", "a");
    if (fout == NULL) printf("Writing results fails.\n");
    else {
        fprintf(fout, "%s,%i,%i,%i,%f,%f\n", filename, rowA, colA, nnzA, wall_ms, gflops);
        fclose(fout);
    }
    if ((rc = hipMemcpy(y, d_y, (size_t)rowA * sizeof(val_t), hipMemcpyDeviceToHost))) die("hipMemcpy y", rc);
    (void)hipFree(d_x); (void)hipFree(d_y);
    tilespmv_plan_destroy(plan);
}

}  // extern "C"
