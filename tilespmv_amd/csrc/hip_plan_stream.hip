// hip_plan_stream.hip — second-generation ("unit stream") layout builder of the plan (hip_plan.h), in stages.
#include <array>
#include <mutex>
#include <set>

#include "hip_plan_internal.h"
#include "plan_tile_ops.h"
#include "hip_plan_device.h"

// ------------------------------------------------------------------------------------------------
// Second-generation layout builder (hip_plan.h "unit stream").
// ------------------------------------------------------------------------------------------------
namespace {

struct RowCount { int nunits, ncoo, nheavy, ndense; long long hval, hidx; long long cost; };

// One tile-row's counts: the sum of its tiles' (plan_tile_ops.h tile_count) plus, in pooled plans, its pool's windows.  `scratch`: room for the tile-row's stored nonzeros (pooled plans)
inline RowCount count_row(const Tile_matrix *T, int bi, int rowlen, int tilen, int colA, bool coo_in_tile, bool dense_mfma, int csr_form, bool absorb, int coo_cost, const long long *hyb_off, std::vector<PoolEnt> &scratch, long long *pool_stat = nullptr)
{
    RowCount c{0, 0, 0, 0, 0, 0, 0};
    for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
        const TileCount k = tile_count(T, t, rowlen, tilen, colA, coo_in_tile, dense_mfma, csr_form, absorb && csr_form < 2, T->tile_ptr[bi], T->tile_ptr[bi + 1]);
        c.nunits += k.nunits; c.ncoo += k.ncoo; c.nheavy += k.nheavy; c.ndense += k.ndense; c.hval += k.hval; c.hidx += k.hidx;
    }
    if (csr_form >= 2) {
        scratch.resize((size_t)std::max<long long>(1, pool_row_capacity(T, bi)));
        int nu, nc, nl;
        pool_row_count(T, bi, rowlen, coo_in_tile, hyb_off, csr_form == 3 ? POOL_WIDE_WINDOW : 16u, scratch.data(), &nu, &nc, &nl);
        c.nunits += nu; c.ncoo += nc;
        if (pool_stat) { pool_stat[0] += nu; pool_stat[1] += nl; }
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


// The builder, one stage per method.  Every stage reads the members earlier stages filled and, in layout-digest builds
// (plan->dry), hashes what it produced into plan->stage_digest[stage]: tests/test_plan_stages.py pins which knob may change which stage.
//   COUNT    per tile-row: units / entries / whole tiles / dense tiles it will emit, cost; prefix sums
//   CHOOSE   strip size, entry mode, strips per workgroup, ordered adds, brick order: pure function of the counts and the knobs
//   CUT      strips of whole tile-rows and pieces of split tile-rows -> task records
//   EMIT     unit descriptors + values, entry triples, whole-tile and dense-tile payload, in tile order
//   ORDER    task order: linear, or brick order (stencil-like shards) + the workgroups' x windows
//   ENCODE   final HBM form of the units: value groups per task, 12-B descriptors or 4-B words + pattern dictionary
//   ENTRIES  merged, column-ordered, packed entry lists per wavefront / workgroup
//   FINISH   remaining uploads, cache-policy flags, byte model
struct StreamBuilder {
    tilespmv_plan *plan; const Knobs &K; const Tile_matrix *T; const int rowA, colA, tr0, tr1; const bool coo_in_tile, dense_mfma;
    const std::vector<long long> &hyb_off;
    const long long *hyb_ptr() const { return hyb_off.empty() ? nullptr : hyb_off.data(); }
    DevShard DS{};
    DevCounts dcnt, dcnt_alt;
    uint4 *d_udesc = nullptr, *d_ucol = nullptr; uint2 *d_urow = nullptr; val_t *d_uval = nullptr;   // EMIT's unit records (scratch: ENCODE writes their final form into the plan's arena)
    hvec<unsigned> h_uw0;   // word 0 of every emitted unit (brick order only)
    std::vector<FixRow> &fix; int &npartial;
    DevStream &S;
    // device mode (hip_plan_device.h; tilespmv_plan_create_from_csr): T is then a host copy of the tile LIST only (tile_ptr, tile_columnidx, Format); everything else of the
    // tiled matrix stays on the device, and the stages that touch every nonzero run there
    const DevTile *DT = nullptr;
    int rc = 0;
    // COUNT
    bool csr_split = true, pooled = false, wide = false, absorb = false, derive = false; int csr_form = 1, target_in = 0, split_above_in = 0, tilem = 0, tilen = 0, ntr = 0, sv = 0;
    hvec<RowCount> rc_;   // (hvec: huge-page advice on the large per-tile-row arrays, host_util.h)
    hvec<long long> pu, pc, ph, phv, phi, pd;
    long long NU = 0, NC = 0, NH = 0, NHV = 0, NHI = 0, ND = 0;
    // CHOOSE
    bool fix_inline_on = true, entry_heavy = false, entry_dominated = false, wave_coo = false, coo_ordered = false, brick = false;
    long long total_cost = 0, est_wgs = 0;
    int target = 0, entry_mode = 0, wg_strips = 16, xs1 = 0, xs2 = 0, max_strip_rows = STRIP_MAX_ROWS;
    int panel_shift = 0, x_panels = 1;                          // column panels of the merged entry lists (hip_plan.h DevStream::x_panels)
    // CUT
    std::vector<STask> tasks;
    std::vector<Task> htasks;
    std::vector<FixRow> ifix, fix_late;   // split rows summed inside the unit kernel / by k_fixup_split after all passes
    std::vector<DenseRow> drows;
    std::vector<unsigned char> row_k, row_split;
    int npartial0 = 0;
    // EMIT
    std::vector<uint4> h_udesc;
    std::vector<uint2> h_urow;            // pooled plans: row nibbles of every unit
    std::vector<uint4> h_ucol;            // wide pooled plans: column-offset bytes of every unit
    val_t *h_uval = nullptr, *h_cval = nullptr, *h_hval = nullptr, *h_dval = nullptr;
    unsigned char *h_hidx = nullptr;
    std::vector<int> h_ccol, h_dcb;
    std::vector<unsigned char> h_crow;
    std::vector<uint2> h_hdesc;
    // ORDER
    std::vector<uint4> h_udesc_cb;
    // ENCODE / ENTRIES
    long long NUP = 0, n_rec = 0, n_chunk = 0, n_groups = 0, panel_rmw_rows = 0;
    bool pool_dict = false;   // pooled plan with 8-B descriptors + pattern dictionary
    long long desc_bytes() const { return S.cb_bits > 0 ? 4 : wide ? 28 : pooled ? (pool_dict ? 8 : 20) : 12; }
    std::vector<long long> old_begin;

    StreamBuilder(tilespmv_plan *plan_, const Knobs &K_, const Tile_matrix *T_, int rowA_, int colA_, int tr0_, int tr1_, bool coo_in_tile_, bool dense_mfma_,
                  const std::vector<long long> &hyb_off_, std::vector<FixRow> &fix_, int &npartial_, const DevTile *DT_)
        : plan(plan_), K(K_), T(T_), rowA(rowA_), colA(colA_), tr0(tr0_), tr1(tr1_), coo_in_tile(coo_in_tile_), dense_mfma(dense_mfma_), hyb_off(hyb_off_), fix(fix_), npartial(npartial_), S(plan_->st), DT(DT_) {}
    ~StreamBuilder() { release(); dcnt.release(); dcnt_alt.release(); for (void *q : {(void *)d_udesc, (void *)d_urow, (void *)d_uval, (void *)d_ucol}) if (q) (void)hipFree(q); }
    // The staging arrays of a GB-sized plan take tens of milliseconds to give back (munmap of 0.7 GB each: 70-80 ms of the 320 ms config 4's plan build took): a detached
    // thread does it while the builder goes on.  Small arrays are freed in place.
    static void free_later(std::vector<void *> ptrs, size_t bytes_hint)
    {
        ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), (void *)nullptr), ptrs.end());
        if (ptrs.empty()) return;
        if (bytes_hint < ((size_t)64 << 20)) { for (void *q : ptrs) free(q); return; }
        std::thread([ptrs]() { for (void *q : ptrs) free(q); }).detach();
    }
    void release()
    {
        free_later({h_uval, h_cval, h_hval, h_hidx, h_dval}, (size_t)(NU * 16 + NC + NHV + ND * 256) * sizeof(val_t));
        h_uval = h_cval = h_hval = h_dval = nullptr; h_hidx = nullptr;
    }

    // ---- stage digests (layout-digest builds only): FNV-1a-64 over (element count, bytes) of what the stage produced
    struct Hash {
        unsigned long long h = 1469598103934665603ull;
        void bytes(const void *p, size_t len) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < len; i++) { h ^= b[i]; h *= 1099511628211ull; } }
        template <class V, class A> void vec(const std::vector<V, A> &v) { arr(v.data(), v.size()); }
        template <class V> void arr(const V *p, size_t n) { const unsigned long long cnt = n; bytes(&cnt, 8); if (n) bytes(p, n * sizeof(V)); }
        void num(long long v) { bytes(&v, 8); }
    };
    bool hashing() const { return plan->dry; }
    void stage_done(int stage, const Hash &h) { plan->stage_digest[stage] = h.h; }

    void count();
    void choose();
    void cut();
    void emit();
    void order();
    void encode();
    void encode_device();
    void entries();
    void finish(long long &n_tasks, long long &model_bytes);
};

void StreamBuilder::count()
{
    target_in = K.strip_cost; split_above_in = K.split_above;
    tilem = T->tilem; tilen = T->tilen; ntr = std::max(0, tr1 - tr0); sv = (int)sizeof(val_t);
    // ---- what CSR tiles become: 0 whole tiles in their own pass, 1 ELL-style split (w padded units + list entries), 2 pooled units (hip_plan.h), unset: 1 or 2 by the bytes
    // the two forms put into the streams.  Both are counted (a shard without CSR tiles is not: nothing to choose); the pooled form is taken when its streams are at least
    // 5 % smaller than the split form's WITH 4-byte dictionary descriptors (whether the dictionary applies is only known once the units exist: the split form gets the benefit
    // of the doubt, so stencil-like shards with a few CSR tiles — KKT, unaligned grids — keep their 4-byte descriptors).
    csr_form = K.csr_split < 0 ? 1 : std::min(K.csr_split, 3);
    absorb = K.absorb != 0 && coo_in_tile;   // (classic forms only: the per-tile functions ignore it in pooled plans)
    derive = absorb && K.absorb != 2;        // derived units (plan_tile_ops.h): absorb = 2 keeps every unit gathering
    const int t_begin = T->tile_ptr[tr0], t_end = T->tile_ptr[tr1];
    if (DT) {
        DS = DevShard{DT, tr0, tr1, t_begin, t_end, coo_in_tile, dense_mfma, 0, 0, absorb, derive};
        const long long idx[2] = {t_begin, t_end}; int v[2] = {0, 0};
        if (dev_fetch_ints(DT->T.blknnz, idx, 2, v) != 0) { rc = -3; return; }
        DS.stored0 = v[0]; DS.stored = (long long)v[1] - v[0];
    }
    long long pool_units_of[4] = {0, 0, 0, 0}, pool_lines_of[4] = {0, 0, 0, 0};   // per form: units made of pooled windows, 128-byte lines of x their gathers touch
    auto count_all = [&](int form, hvec<RowCount> &out) {
        out.assign((size_t)ntr, RowCount{0, 0, 0, 0, 0, 0, 0});
        if (DT) {   // one thread per tile (per tile-row for the pooled windows) on the device: the same per-tile functions (plan_tile_ops.h)
            hvec<int> k3;   // (units, entries, dense tiles) of every tile-row
            DevCounts &C = form == csr_form ? dcnt : dcnt_alt;
            if (dev_count(DS, form, &C, k3, &pool_units_of[form], &pool_lines_of[form]) != 0) { rc = -3; return; }
            parallel_chunks(ntr, 1 << 16, [&](int64_t b, int64_t e, int) {
                for (int64_t i = b; i < e; i++) {
                    RowCount c{k3[3 * (size_t)i], k3[3 * (size_t)i + 1], 0, k3[3 * (size_t)i + 2], 0, 0, 0};
                    c.cost = 16LL * c.nunits + (long long)K.coo_cost * c.ncoo + 64LL * c.ndense + 8;
                    out[(size_t)i] = c;
                }
            });
            return;
        }
        std::mutex stat_mutex;
        parallel_chunks(ntr, 1024, [&](int64_t b, int64_t e, int) {
            std::vector<PoolEnt> scratch;
            long long st[2] = {0, 0};
            for (int64_t i = b; i < e; i++) out[i] = count_row(T, tr0 + (int)i, tile_rowlen(tr0 + (int)i, tilem, rowA), tilen, colA, coo_in_tile, dense_mfma, form, absorb, K.coo_cost, hyb_ptr(), scratch, st);
            std::lock_guard<std::mutex> lk(stat_mutex);
            pool_units_of[form & 3] += st[0]; pool_lines_of[form & 3] += st[1];
        });
    };
    count_all(csr_form, rc_);
    if (rc) return;
    // (a caller who asks for a launch form that exists for the classic units only — column panels / slices, pacing, 512-thread workgroups, LDS x windows, a forced dictionary —
    //  gets the classic units)
    const bool classic_asked = K.x_panel_kb > 0 || K.x_slice_passes > 0 || K.wg_strips == 32 || K.desc_dict == 1;
    // (nor is the second count worth its time where CSR tiles hold less than 3 % of the shard's stored values — the KKT stand-in: under 1 %; the band matrix's 5 % is worth it —: the pooled form pays 8-16 bytes more per
    //  unit on everything else and cannot come out 5 % ahead)
    long long csr_vals, all_vals;
    long long e16 = -1;   // list entries the 16-column pooled form would leave (-1: not counted)
    if (DT) {
        const long long idx[2] = {t_begin, t_end}; int v[2] = {0, 0};
        if (dev_fetch_ints(DT->T.csr_offset, idx, 2, v) != 0) { rc = -3; return; }
        csr_vals = (long long)v[1] - v[0]; all_vals = DS.stored;
    } else { csr_vals = (long long)T->csr_offset[t_end] - T->csr_offset[t_begin]; all_vals = (long long)T->blknnz[t_end] - T->blknnz[t_begin]; }
    // (round 6: ... or where the classic form leaves a tenth of the shard on entry lists — stencils on grids whose lines are not a multiple of 16 long: their off-diagonals cross the
    //  tile boundaries at a different row in every tile-row, the pieces are COO tiles of 1-12 entries, and a pooled unit's window starts at any column: 7-point 250^3 0.2505 -> 0.2235 ms,
    //  5-point 4090^2 0.1835 -> 0.1756; irregular shards are counted too and keep the classic form by the same byte rule)
    long long e_classic = 0;
    for (int i = 0; i < ntr; i++) e_classic += rc_[i].ncoo;
    const bool consider_pooled = K.csr_split < 0 && !classic_asked && ((csr_vals > 0 && csr_vals * 33 >= all_vals) || (coo_in_tile && e_classic * 10 >= all_vals));
    if (consider_pooled) {
        hvec<RowCount> alt;
        count_all(2, alt);
        if (rc) return;
        long long u1 = 0, e1 = 0, u2 = 0, e2 = 0;
        for (int i = 0; i < ntr; i++) { u1 += rc_[i].nunits; e1 += rc_[i].ncoo; u2 += alt[i].nunits; e2 += alt[i].ncoo; }
        e16 = e2;
        // would the split form get 4-byte dictionary descriptors?  Its units' column patterns on a sample of tiles (ELL slots exactly; of a CSR tile the pattern of its first
        // unit: the first column nibble of every row): more than the dictionary holds on the sample -> 12-byte descriptors for certain
        int desc_split = 4;
        if (K.desc_dict == 0) desc_split = 12;
        else {
            std::unordered_set<unsigned long long> pats;
            const int nt = t_end - t_begin, step = std::max(1, nt / 16384);
            // tiles of a partial last tile-row have fewer than 16 rows: their row pointer advances by rowlen and their ELL slots are rowlen apart (the sample read 16)
            const int last_rowlen = tile_rowlen(tilem - 1, tilem, rowA), last_first_tile = tilem > 0 ? T->tile_ptr[tilem - 1] : 0;
            if (DT) {   // the same sample, taken by a kernel
                std::vector<unsigned long long> got;
                if (dev_pattern_sample(DS, step, last_first_tile, last_rowlen, got) != 0) { rc = -3; return; }
                pats.insert(got.begin(), got.end());
            } else
            for (int t = t_begin; t < t_end && pats.size() <= ((size_t)1 << DICT_MAX_BITS); t += step) {
                const int fmt = T->Format[t];
                const int rl = t >= last_first_tile ? last_rowlen : 16;
                if (fmt == TILESPMV_FMT_ELL) {
                    const int off = T->ell_offset[t], w = T->tilewidth[t];
                    for (int sl = 0; sl < w; sl++) { unsigned long long nibs = 0; for (int r = 0; r < rl; r++) nibs |= (unsigned long long)nib(T->ell_compressedIdx, (long long)off + sl * rl + r) << (60 - 4 * r); pats.insert(nibs); }
                } else if (fmt == TILESPMV_FMT_CSR) {
                    const int off = T->csr_offset[t], stored = T->blknnz[t + 1] - T->blknnz[t];
                    const unsigned char *ptr = T->Blockcsr_Ptr + T->csrptr_offset[t];
                    unsigned long long nibs = 0;
                    for (int r = 0; r < rl; r++) { const int k0 = ptr[r], k1 = r == rl - 1 ? stored : ptr[r + 1]; if (k1 > k0) nibs |= (unsigned long long)nib(T->csr_compressedIdx, (long long)off + k0) << (60 - 4 * r); }
                    pats.insert(nibs);
                }
            }
            if (pats.size() > ((size_t)1 << DICT_MAX_BITS)) desc_split = 12;
        }
        const long long xy_b = ((long long)colA + 16LL * ntr) * sv;   // x once, y once: part of what either form moves
        const long long split_b = u1 * (desc_split + 16LL * sv) + e1 * (sv + 5LL) + xy_b, pooled_b = u2 * (20 + 16LL * sv) + e2 * (sv + 5LL) + xy_b;
        // Calibrated on the population (scripts/archive/rounds/r5_form_study.py, profiles/r05_form_study.txt): the pooled kernel's time follows its bytes with about 8 % on top of the
        // classic kernel's at equal bytes, so it is taken where one SpMV moves at least 5 % fewer bytes — and only where units carry the shard (an entry-dominated shard
        // lives in its entry lists) and the shard would not get column panels (scattered entries over an x of >= 12 MB: the panel / slice launches exist for the
        // classic kernel; band + random fill loses 23 % without them)
        const bool would_panel = (long long)K.coo_cost * e2 * 2 > 16LL * u2 + (long long)K.coo_cost * e2 && (long long)colA * sv >= (12ll << 20);
        bool take = pooled_b * 100 <= split_b * 95 && 16 * u2 >= e2 && !would_panel;
        // the second way in (round 6): a regular shard (its split form would get the dictionary) a tenth of which sat on entry lists, and the pooled windows take at least half of those
        // entries off the lists — stencils on grids whose lines are not a multiple of 16 long.  There the classic kernel's time does not follow its bytes (its entry phase is the slow part):
        // any byte saving will do, with the pooled units priced at the 4-byte dictionary words such shards get.  (Irregular shards fail the half: power-law 28.8 -> 25.6 M entries, road
        // network 20.3 -> 13.6 M, and are 8-23 % slower pooled.)
        const long long pooled_reg_b = u2 * (4 + 16LL * sv) + e2 * (sv + 5LL) + xy_b;
        const bool take_regular = !take && coo_in_tile && e_classic * 10 >= all_vals && desc_split == 4 && e2 * 2 <= e1 && pooled_reg_b <= split_b && 16 * u2 >= e2 && !would_panel;
        if (take || take_regular) { rc_.swap(alt); csr_form = 2; std::swap(dcnt, dcnt_alt); dcnt.csr_form = 2; }
        dcnt_alt.release();
        if (getenv("TILESPMV_PLAN_VERBOSE")) fprintf(stderr, "tilespmv: CSR tiles: split form (%d-byte descriptors) %lld units + %lld entries = %.1f MB, pooled form %lld units + %lld entries = %.1f MB -> %s\n",
                                                     desc_split, u1, e1, split_b / 1e6, u2, e2, pooled_b / 1e6, csr_form == 2 ? (take ? "pooled" : "pooled (regular shard, entries halved)") : "split");
    }
    // ---- wide pooled units (hip_plan.h; csr_form 3): windows of 256 columns take what 16-column windows leave on the entry lists — where the nonzeros are dense enough inside those
    // windows that a unit's 16 gathers still touch few lines of x (window-shuffled FEM / shell meshes: 1.5-3 lines per unit) it wins 5-20 %; where every slot sits on a line of its own
    // (tetrahedral / 2-D meshes shuffled over thousands of nodes, web graphs) the column-ordered entry lists of a whole workgroup share lines better and the wide form loses up to 30 %
    // (profiles/r05_wide_windows.txt).  Counted whenever the 16-column pooled form was, taken by the rule below.
    if (consider_pooled) {
        long long ub = 0, eb = 0;
        for (int i = 0; i < ntr; i++) { ub += rc_[i].nunits; eb += rc_[i].ncoo; }
        hvec<RowCount> alt;
        count_all(3, alt);
        if (rc) return;
        long long u3 = 0, e3 = 0;
        for (int i = 0; i < ntr; i++) { u3 += alt[i].nunits; e3 += alt[i].ncoo; }
        const double lines3 = (double)pool_lines_of[3] / (double)std::max(1LL, pool_units_of[3]);
        const long long xy_b = ((long long)colA + 16LL * ntr) * sv;
        const long long best_b = ub * ((csr_form == 2 ? 20 : 12) + 16LL * sv) + eb * (sv + 5LL) + xy_b, wide_b = u3 * (28 + 16LL * sv) + e3 * (sv + 5LL) + xy_b;
        const long long moved = (e16 >= 0 ? std::min(eb, e16) : eb) - e3;   // nonzeros that leave the entry lists — measured against the 16-column windows too, chosen or not (a natural-order mesh whose
                                                                             // 16-column form narrowly missed its own bar must not get 256-column windows it has no use for)
        // (column panels exist for the classic units: a shard that would still be entry-dominated with wide windows keeps them — circuit-like shards, where wide units take 80 % of the
        //  nonzeros, do not: 0.081 -> 0.074 ms on the 4 M-row one)
        const bool would_panel = (long long)K.coo_cost * e3 * 2 > 16LL * u3 + (long long)K.coo_cost * e3 && 3 * e3 > 16 * u3 && (long long)colA * sv >= (12ll << 20);
        const bool take = lines3 <= POOL_WIDE_MAX_LINES && moved * 25 >= (16 * u3 + e3) && wide_b * 100 <= best_b * 103 && 16 * u3 >= e3 && !would_panel;
        if (getenv("TILESPMV_PLAN_VERBOSE")) fprintf(stderr, "tilespmv: wide windows: %lld units + %lld entries = %.1f MB (chosen so far: %lld + %lld = %.1f MB), %.2f lines of x per pooled unit, %.1f %% of the nonzeros leave the lists -> %s\n",
                                                     u3, e3, wide_b / 1e6, ub, eb, best_b / 1e6, lines3, 100.0 * moved / std::max(1LL, 16 * u3 + e3), take ? "wide" : "kept");
        if (take) { rc_.swap(alt); csr_form = 3; std::swap(dcnt, dcnt_alt); dcnt.csr_form = 3; }
        dcnt_alt.release();
    }
    csr_split = csr_form != 0; pooled = csr_form >= 2; wide = csr_form == 3;
    {   // (six arrays of ntr + 1 prefixes: first touched side by side — a million tile-rows are 50 MB of fresh pages)
        hvec<long long> *six[6] = {&pu, &pc, &ph, &phv, &phi, &pd};
        parallel_chunks(6, 1, [&](int64_t b, int64_t e, int) { for (int64_t q = b; q < e; q++) six[q]->assign((size_t)ntr + 1, 0); });
    }
    for (int i = 0; i < ntr; i++) pd[i + 1] = pd[i] + rc_[i].ndense;
    ND = pd[ntr];
    for (int i = 0; i < ntr; i++) {
        pu[i + 1] = pu[i] + rc_[i].nunits; pc[i + 1] = pc[i] + rc_[i].ncoo; ph[i + 1] = ph[i] + rc_[i].nheavy;
        phv[i + 1] = phv[i] + rc_[i].hval; phi[i + 1] = phi[i] + rc_[i].hidx;
    }
    NU = pu[ntr]; NC = pc[ntr]; NH = ph[ntr]; NHV = phv[ntr]; NHI = phi[ntr];
    if (tilen > (1 << UNIT_FLAG_SHIFT)) { fprintf(stderr, "tilespmv: more than 2^24 column blocks: use TILESPMV_KERNEL=1\n"); rc = -2; return; }
    if (NU > INT32_MAX || NC > INT32_MAX || NH > INT32_MAX) { fprintf(stderr, "tilespmv: shard too large for 32-bit unit ids\n"); rc = -2; return; }
    if (hashing()) { Hash h; h.vec(rc_); for (long long v : {NU, NC, NH, NHV, NHI, ND, (long long)csr_form}) h.num(v); stage_done(TILESPMV_STAGE_COUNT_ROWS, h); }
}

void StreamBuilder::choose()
{
    // ---- strips (<= STRIP_MAX_ROWS whole tile-rows up to the cost target) for the unit kernel, one
    // heavy task per tile-row that owns heavy tiles, and pieces of very long tile-rows (all three
    // kinds of pieces write partial[] slots that k_fixup_split adds up in a fixed order).
    fix_inline_on = K.fix_inline != 0;
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
    entry_heavy = NC >= 5LL * ntr;   // (5 since round 3: an unaligned 7-point grid — 6 one-entry COO tiles per tile-row — runs 5 % faster with the workgroup entry mode; 4 per tile-row, the unaligned 5-point grid, does not)
    total_cost = 0;
    for (int i = 0; i < ntr; i++) total_cost += rc_[i].cost;
    // tilespmv_plan_spmm: k_units_mv walks a strip's entries with its 16 lanes, tile-row by tile-row; where entries are most
    // of the work (webbase-like: nvec 2 took 0.11 ms against 0.013 ms for one SpMV) one SpMV per right-hand side is faster
    entry_dominated = (long long)K.coo_cost * NC * 2 > total_cost;
    target = target_in;
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
    // pooled plans (strips of at most 4 tile-rows): light tile-rows want FULL strips — tet150: 15 units per tile-row, 0.1025 ms at 400 (1-2 rows per strip, 208 k tasks), 0.0954 at 800
    // (4 rows, 53 k tasks) — and heavy ones want to stay whole: fem12_20's 290-unit tile-rows were cut into pieces at the 6 x 400 split threshold (0.0542 -> 0.0513 ms unsplit);
    // fem3_68 / fem3s64 / tri2200 do not move (profiles/r05_pool_strip_cost.txt)
    // (only where 800 still leaves >= 4 workgroups per CU: the scircuit stand-in — 172 workgroups at 800 — runs 7.6 us against 6.6-7.0 at 400)
    if (target_in <= 0 && pooled && target < 800 && total_cost / (16LL * 800) >= 1024) target = 800;
    target = std::max(32, target);
    est_wgs = total_cost / (16LL * target) + 1;
    const int wave_coo_env = K.entry_mode;
    entry_mode = wave_coo_env >= 0 ? std::min(2, wave_coo_env) : (!entry_heavy ? 0 : est_wgs < 768 ? 1 : 2);
    wave_coo = entry_mode != 0;
    plan->entry_mode = entry_mode;
    if (getenv("TILESPMV_PLAN_VERBOSE")) {
        long long rows_over16 = 0;   // tile-rows with more entries than travel with a strip's prologue
        for (int i = 0; i < ntr; i++) rows_over16 += rc_[i].ncoo > 16;
        fprintf(stderr, "tilespmv: choose: %d tile-rows, %lld units + %lld entries (%.1f per tile-row, %lld tile-rows with more than 16), cost %lld (%.0f per tile-row) -> strips of %d, ~%lld workgroups, entry mode %d\n",
                ntr, NU, NC, ntr ? (double)NC / ntr : 0.0, rows_over16, total_cost, ntr ? (double)total_cost / ntr : 0.0, target, est_wgs, entry_mode);
    }
    // strips per workgroup: 32 (512 threads) only on request and only with the workgroup entry mode — twice as many tile-rows share
    // one column-ordered list (power-law 8 M rows: 0.204 -> 0.152 distinct 128-B x lines per entry) at the same 6 waves per SIMD, but
    // it measures slower everywhere (power-law 8 M 0.1038 -> 0.1072 ms, webbase-like 13.1 -> 13.9 us, KKT fp64 equal): default 16
    wg_strips = (entry_mode == 2 && K.wg_strips == 32 && !pooled) ? 32 : 16;   // (pooled plans: 256-thread workgroups only)
    plan->wg_strips = wg_strips;
    // Workgroup mode: the four wavefronts add into shared slabs.  Taking turns (4 barriers per trip) fixes the order of the
    // additions -> bit-reproducible sums; free or a gain on large grids (fewer LDS conflicts: power-law 8 M rows 0.122 ->
    // 0.118 ms), +8 % on mid-size ones (webbase-like 14.2 -> 15.3 us), which therefore add unordered unless
    // TILESPMV_COO_ORDERED=1 asks for reproducible bits.  Modes 0 and 1 are always ordered (one wavefront per slab).
    const int ordered_env = K.entry_ordered;
    coo_ordered = ordered_env >= 0 ? ordered_env != 0 : est_wgs >= 2048;
    // ---- brick order (stencil-like shards): the grid strides of the shard are detected from its tile pattern, strips stay inside
    // one grid line, and after the cut the strips are regrouped so that the 16 strips of a workgroup — and the neighbouring
    // workgroups of an XCD window — form a brick of the grid instead of a run of one grid line: the x segments a tile-row shares
    // with its neighbours in the other two directions are then wanted at about the same time by one CU / one XCD, and hit in L1 /
    // L2 instead of being fetched again (nlpkkt160 stand-in: 3.07 -> 2.6-2.7 GB per launch at the fabric in fp64, 1.81 -> 1.56 GB in fp32;
    // time -2.5 ... -6 % in fp32, inside the matrix's 10 % placement spread in fp64: DESIGN.md S6.9).
    //   x_window  -1 (default): brick order on large 3-D shards   0: off   1 / 2: brick order wherever strides are found
    //   (rounds 3-5 also had 1 = the workgroup's x segments staged once in LDS: cut another ~0.4 GB but ran 25 % slower — profiles/r03_xwindow_and_map.txt; retired in round 6)
    xs1 = K.x_stride1 > 0 ? K.x_stride1 : 0; xs2 = K.x_stride2 > 0 ? K.x_stride2 : 0;
    brick = K.x_window != 0 && wg_strips == 16 && (K.x_window > 0 || est_wgs >= 2048);
    if (brick && xs1 == 0) detect_strides(T, tr0, tr1, csr_split, dense_mfma, &xs1, &xs2);
    if (xs1 < 2 || (K.x_window < 0 && xs2 == 0)) brick = false;   // (2-D grids: measured neutral on the 5-point 4096^2 case)
    // (strips of at most 4 tile-rows in brick plans: nlpkkt160 stand-in fp64 0.418 -> 0.414 ms, fp32 0.252 -> 0.250 in one process; 2 rows: KKT 0.408 but 7-pt 256^3 +5 %)
    max_strip_rows = pooled ? POOL_STRIP_ROWS : brick ? (K.brick_rows > 0 ? std::min(K.brick_rows, STRIP_MAX_ROWS) : 4) : STRIP_MAX_ROWS;
    if (brick && !K.xcd_from_caller) plan->xcd_chunk = 8;   // bricks are compact: smaller XCD windows keep an XCD's resident set together
    // ---- column panels of the entry lists (hip_plan.h DevStream::x_panels): scattered, entry-dominated shards whose x is several times an XCD's L2
    {
        const long long x_bytes = (long long)colA * sv;
        // rule: shards whose work is mostly scattered entries and whose x is several XCD L2s (>= 12 MB) record 2-MB panels; whether a panelled launch pays is then
        // decided by timing (plan_create_one) — it does on uniform-random-like shards (8 M rows: 1.00 -> 0.80 ms) and does not where most entries sit near the diagonal
        const int kb = K.x_panel_kb >= 0 ? K.x_panel_kb : ((entry_dominated && x_bytes >= (12ll << 20)) ? 2048 : 0);
        x_panels = 1; panel_shift = 0;
        if (kb > 0 && entry_mode == 2 && wg_strips == 16 && !pooled && x_bytes > 1024LL * kb) {   // (the panel / slice kernels add into 8-row slabs: classic plans)
            long long cols = std::max<long long>(1024, 1024LL * kb / sv);
            while ((2ll << panel_shift) <= cols) panel_shift++;
            while ((((long long)colA - 1) >> panel_shift) + 1 > 64) panel_shift++;     // at most 64 passes
            x_panels = (int)((((long long)colA - 1) >> panel_shift) + 1);
            if (x_panels <= 1) { x_panels = 1; panel_shift = 0; }
        }
    }
    if (hashing()) {
        Hash h;
        h.num(x_panels); h.num(panel_shift);
        for (long long v : {(long long)target, (long long)entry_mode, (long long)wg_strips, (long long)coo_ordered, (long long)xs1, (long long)xs2, (long long)brick, (long long)max_strip_rows}) h.num(v);
        stage_done(TILESPMV_STAGE_CHOOSE, h);
    }
}

void StreamBuilder::cut()
{
    row_k.assign((size_t)ntr, 0); row_split.assign((size_t)ntr, 0);
    npartial0 = npartial;
    {
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
                j++;
            }
            k.nrows = j - i;
            k.unit_end = (int)pu[j]; k.coo_end = (int)pc[j];
            tasks.push_back(k);
            i = j;
        }

    }
    if (hashing()) { Hash h; h.vec(tasks); h.vec(htasks); h.vec(ifix); h.vec(fix_late); h.vec(drows); h.vec(row_k); h.vec(row_split); h.num(npartial); stage_done(TILESPMV_STAGE_CUT, h); }
}

void StreamBuilder::emit()
{
    if (DT) {
        // device mode: the unit records go to scratch arrays on the device (ENCODE gives them their final form), list entries and dense tiles straight into the plan's arena
        plan->size_hint = (size_t)(NU * (12 + (wide ? 16 : pooled ? 8 : 0) + 16LL * sv) + NC * (2LL * sv + 13) + ND * (4 + 256LL * sv) + (long long)tasks.size() * 40);
        rc |= plan->reserve((size_t)NC, &S.cval); rc |= plan->reserve((size_t)NC, &S.ccol); rc |= plan->reserve((size_t)NC, &S.crow);
        rc |= plan->reserve((size_t)ND, &plan->dn.cb); rc |= plan->reserve((size_t)ND * 256, &plan->dn.val);
        auto scratch = [&](auto **q, size_t n) {
            const size_t bytes = std::max<size_t>(n, 1) * sizeof(**q) + 256;
            hipError_t e = hipMalloc((void **)q, bytes);
            if (e == hipSuccess) e = hipMemsetAsync(*q, 0, bytes, 0);
            if (e != hipSuccess) { fprintf(stderr, "tilespmv: device plan build: %zu MB of scratch: %s\n", bytes >> 20, hipGetErrorString(e)); (void)hipGetLastError(); rc = -3; }
        };
        scratch(&d_udesc, (size_t)NU); scratch(&d_uval, (size_t)NU * 16);
        if (pooled && !wide) scratch(&d_urow, (size_t)NU);
        if (wide) scratch(&d_ucol, (size_t)NU);
        if (rc) return;
        const EmitOut O{d_udesc, d_urow, d_uval, d_ucol, const_cast<val_t *>(S.cval), const_cast<int *>(S.ccol), const_cast<unsigned char *>(S.crow), const_cast<int *>(plan->dn.cb), const_cast<val_t *>(plan->dn.val)};
        if (dev_emit(DS, dcnt, pu, pc, pd, row_k, row_split, NU, O) != 0) rc = -3;
        dcnt.release();
        return;
    }
    // ---- fill
    h_udesc.assign((size_t)NU, make_uint4(0u, 0u, 0u, 0u));
    h_urow.assign(pooled && !wide ? (size_t)NU : 0, make_uint2(0x01234567u, 0x89ABCDEFu));   // (units that keep one row per lane: identity)
    h_ucol.assign(wide ? (size_t)NU : 0, make_uint4(0u, 0u, 0u, 0u));                       // wide pooled plans: column-offset bytes
    h_uval = zalloc<val_t>((size_t)NU * 16);
    h_cval = zalloc<val_t>((size_t)NC);
    h_ccol.assign((size_t)NC, 0);
    h_crow.assign((size_t)NC, 0);
    h_hdesc.assign((size_t)NH, make_uint2(0u, 0u));
    h_hval = zalloc<val_t>((size_t)NHV);
    h_hidx = zalloc<unsigned char>((size_t)NHI + 16);
    h_dcb.assign((size_t)ND, 0);
    h_dval = zalloc<val_t>((size_t)ND * 256);
    const EmitOut O{h_udesc.data(), h_urow.data(), h_uval, h_ucol.data(), h_cval, h_ccol.data(), h_crow.data(), h_dcb.data(), h_dval};
    parallel_chunks(ntr, 256, [&](int64_t b, int64_t e, int) {
        std::vector<PoolEnt> pool;
        for (int64_t i = b; i < e; i++) {
            const int bi = tr0 + (int)i, rowlen = tile_rowlen(bi, tilem, rowA);
            const unsigned kr = row_k[i];
            EmitPos pos{pu[i], pc[i], pd[i]};
            long long h = ph[i], hv = phv[i], hi = phi[i];
            for (int t = T->tile_ptr[bi]; t < T->tile_ptr[bi + 1]; t++) {
                if (T->Format[t] == TILESPMV_FMT_CSR && csr_form == 0) {   // CSR tile as a heavy (whole) tile: the first-generation layout and routine
                    const int cb = T->tile_columnidx[t];
                    Emit em = emit_of(T, t, rowlen, true);
                    repack_tile(T, t, em, rowlen, tile_collen(cb, tilen, colA), 0, h_hval + hv, h_hidx + hi);
                    h_hdesc[(size_t)h] = make_uint2((unsigned)cb, (unsigned)em.fmt | ((unsigned)em.p1 << DESC_P1_SHIFT));
                    h++; hv += em.nv; hi += em.ni;
                } else tile_emit(T, t, rowlen, tilen, colA, coo_in_tile, dense_mfma, csr_form, kr, hyb_ptr(), O, pos, absorb && csr_form < 2, T->tile_ptr[bi], T->tile_ptr[bi + 1], derive && csr_form < 2 && !row_split[i]);   // (plan_tile_ops.h: shared with the device builder)
            }
            if (pooled) {   // the pooled nonzeros of the tile-row: windows of 16 columns -> units, sparse windows -> list entries
                pool.resize((size_t)std::max<long long>(1, pool_row_capacity(T, bi)));
                pool_row_emit(T, bi, rowlen, coo_in_tile, kr, hyb_ptr(), wide ? POOL_WIDE_WINDOW : 16u, pool.data(), O, pos);
            }
            if (!pooled && !row_split[i] && pos.u > pu[i]) { h_udesc[(size_t)pos.u - 1].x |= UNIT_EOR << UNIT_FLAG_SHIFT; h_udesc[(size_t)pos.u - 1].z |= UNIT_EOR << UNIT_FLAG_SHIFT; }
            if (h > ph[i]) h_hdesc[(size_t)h - 1].y |= DESC_EOR;
        }
    });
    if (hashing()) {
        Hash h;
        h.vec(h_udesc); h.vec(h_urow); if (wide) h.vec(h_ucol); h.arr(h_uval, (size_t)NU * 16); h.arr(h_cval, (size_t)NC); h.vec(h_ccol); h.vec(h_crow); h.vec(h_hdesc); h.arr(h_hval, (size_t)NHV); h.arr(h_hidx, (size_t)NHI);
        h.vec(h_dcb); h.arr(h_dval, (size_t)ND * 256);
        stage_done(TILESPMV_STAGE_EMIT, h);
    }
}

void StreamBuilder::order()
{
    // ---- brick order of the strips
    if (brick && !tasks.empty() && DT && dev_fetch_word0(d_udesc, NU, h_uw0) != 0) { rc = -3; return; }
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
            for (int u = k.unit_begin; u < k.unit_end; u++) { const unsigned w0 = DT ? h_uw0[(size_t)u] : h_udesc[(size_t)u].x; out.push_back(pooled ? (int)((w0 & POOL_BASE_MASK) >> 4) : (int)(w0 & 0xFFFFFFu)); }
        };
        struct Shape { int px, py, pz; };
        const Shape shapes3[] = {{1, 4, 4}, {2, 2, 4}, {2, 4, 2}, {4, 2, 2}, {1, 2, 8}, {1, 8, 2}, {4, 4, 1}, {2, 8, 1}, {1, 16, 1}, {16, 1, 1}};
        const Shape shapes2[] = {{4, 4, 1}, {2, 8, 1}, {8, 2, 1}, {1, 16, 1}, {16, 1, 1}};
        const Shape *shapes = xs2 ? shapes3 : shapes2;
        const int nshapes = xs2 ? 10 : 5;
        std::vector<unsigned> best_order;
        double best_avg = 1e30;
        Shape best_shape{16, 1, 1};
        auto sort_for = [&](const Shape &sh, std::vector<unsigned> &order) {
            // one 64-bit key per strip (brick coordinates, then the position inside the brick: 10 bits per field is plenty below 2^30 tile-rows per plane ... checked below), ties by index
            order.resize(nt);
            std::vector<std::pair<unsigned long long, unsigned>> key(nt);
            bool fits = true;
            for (size_t i = 0; i < nt; i++) {
                const unsigned long long k0 = (unsigned long long)(lz[i] / sh.pz), k1 = (unsigned long long)(ly[i] / sh.py), k2 = (unsigned long long)(sx[i] / sh.px);
                const unsigned long long k3 = (unsigned long long)(lz[i] % sh.pz), k4 = (unsigned long long)(ly[i] % sh.py), k5 = (unsigned long long)(sx[i] % sh.px);
                if (k0 >= (1ull << 18) || k1 >= (1ull << 18) || k2 >= (1ull << 16)) fits = false;
                key[i] = {(k0 << 46) | (k1 << 28) | (k2 << 12) | (k3 << 8) | (k4 << 4) | k5, (unsigned)i};
            }
            if (fits) {
                std::sort(key.begin(), key.end());
                for (size_t i = 0; i < nt; i++) order[i] = key[i].second;
                return;
            }
            for (size_t i = 0; i < nt; i++) order[i] = (unsigned)i;
            std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
                const int ka[6] = {lz[a] / sh.pz, ly[a] / sh.py, sx[a] / sh.px, lz[a] % sh.pz, ly[a] % sh.py, sx[a] % sh.px};
                const int kb[6] = {lz[b] / sh.pz, ly[b] / sh.py, sx[b] / sh.px, lz[b] % sh.pz, ly[b] % sh.py, sx[b] % sh.px};
                for (int q = 0; q < 6; q++) if (ka[q] != kb[q]) return ka[q] < kb[q];
                return a < b;
            });
        };
        const size_t nwg = (nt + 15) / 16;
        // the brick shape that needs the fewest window slots on a sample of workgroups: the candidate shapes are sorted and scored side by side (round 5: one after the other this
        // stage took 300 ms of the 0.8 s the KKT stand-in's plan build needs), the choice among them is made in the fixed order of the list
        std::vector<std::vector<unsigned>> orders((size_t)nshapes);
        std::vector<double> avgs((size_t)nshapes, 1e30);
        parallel_chunks(nshapes, 1, [&](int64_t b0, int64_t b1, int) {
            std::vector<int> tmp;
            for (int64_t si = b0; si < b1; si++) {
                sort_for(shapes[si], orders[(size_t)si]);
                long long slots = 0, wgs = 0;
                for (size_t w = nwg / 128; w < nwg; w += std::max<size_t>(1, nwg / 64)) {
                    tmp.clear();
                    for (size_t t = 16 * w; t < std::min(nt, 16 * w + 16); t++) blocks_of(tasks[orders[(size_t)si][t]], tmp);
                    std::sort(tmp.begin(), tmp.end());
                    slots += (long long)(std::unique(tmp.begin(), tmp.end()) - tmp.begin()); wgs++;
                }
                avgs[(size_t)si] = wgs ? (double)slots / (double)wgs : 1e30;
            }
        });
        static const int forced_shape = [] { const char *e = getenv("TILESPMV_BRICK_SHAPE"); return e && *e ? atoi(e) : -1; }();   // (experiment knob, environment only: index into the shape list)
        for (int si = 0; si < nshapes; si++)
            if (forced_shape >= 0 && forced_shape < nshapes ? si == forced_shape : avgs[(size_t)si] < best_avg * 0.98) { best_avg = avgs[(size_t)si]; best_order.swap(orders[(size_t)si]); best_shape = shapes[si]; }
        {
            std::vector<STask> permuted(nt);
            for (size_t i = 0; i < nt; i++) permuted[i] = tasks[best_order[i]];
            tasks.swap(permuted);
        }
        if (getenv("TILESPMV_PLAN_VERBOSE"))
            fprintf(stderr, "tilespmv: brick order: strides %d / %d tile-rows, brick %d x %d x %d strips, %.1f distinct column blocks per workgroup on the sample\n",
                    xs1, xs2, best_shape.px, best_shape.py, best_shape.pz, best_avg);
    } else brick = false;
    plan->info[TILESPMV_INFO_BRICK_ORDER] = brick ? 1 : 0;
    plan->info[TILESPMV_INFO_LIST_ENTRIES] = NC;
    plan->size_hint = (size_t)(NU * (12 + (wide ? 16 : pooled ? 8 : 0) + 16LL * sv) + NC * (2LL * sv + 13) + NHV * sv + NHI + ND * (4 + 256LL * sv) + (long long)tasks.size() * 40);   // estimate of the plan's bytes: picks the block size of upload()
    if (hashing()) { Hash h; h.vec(tasks); h.num(brick); stage_done(TILESPMV_STAGE_ORDER, h); }
}

// ENCODE in device mode: the same final forms, produced from EMIT's device scratch (hip_plan_device.h)
void StreamBuilder::encode_device()
{
    constexpr long long G = UNIT_GROUP;
    auto padded = [&](long long n) { return (n + G - 1) / G * G; };
    std::vector<int4> pair_map(tasks.size());
    old_begin.assign(tasks.size(), 0);
    long long at = 0;
    for (size_t i = 0; i < tasks.size(); i++) {
        STask &k = tasks[i];
        const long long ub = k.unit_begin, n = k.unit_end - ub, nb = at;
        old_begin[i] = ub;
        at += padded(n);
        pair_map[i] = make_int4((int)ub, (int)nb, (int)n, 0);
        if (n > 0) { k.unit_begin = (int)nb; k.unit_end = (int)(nb + n); }
    }
    const double t0 = now_us();
    const long long up0 = plan->info[TILESPMV_INFO_UPLOAD_US];   // (reserve() / upload() below count their own time: the stage's total replaces it at the end, not adds to it)
    S.udict = nullptr; S.cb_bits = 0; S.urow = nullptr; S.ucol = nullptr; S.pooled = pooled ? 1 : 0; S.pdict = nullptr; pool_dict = false;
    void *d_map = nullptr; UDesc *d_packed = nullptr; URow *d_prow = nullptr;
    auto fail = [&](const char *what, hipError_t e) { fprintf(stderr, "tilespmv: device plan build: %s: %s\n", what, hipGetErrorString(e)); (void)hipGetLastError(); rc = -3; };
    hipError_t e = hipMalloc(&d_map, std::max<size_t>(pair_map.size(), 1) * sizeof(int4));
    if (e == hipSuccess && !pair_map.empty()) e = hipMemcpy(d_map, pair_map.data(), pair_map.size() * sizeof(int4), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&d_packed, std::max<long long>(NUP, 1) * sizeof(UDesc) + 256);
    if (e == hipSuccess) e = hipMemsetAsync(d_packed, 0, std::max<long long>(NUP, 1) * sizeof(UDesc) + 256, 0);
    if (e != hipSuccess) fail("scratch for the packed descriptors", e);
    if (wide) rc |= plan->reserve((size_t)NUP, &S.ucol);   // (wide pooled plans: the column-offset bytes are packed straight into the plan's arena)
    if (pooled && !wide && e == hipSuccess) {   // packed row nibbles: scratch (they end up in the plan's arena, or — pooled dictionary plans — in the dictionary)
        e = hipMalloc((void **)&d_prow, std::max<long long>(NUP, 1) * sizeof(URow) + 256);
        if (e == hipSuccess) e = hipMemsetAsync(d_prow, 0, std::max<long long>(NUP, 1) * sizeof(URow) + 256, 0);
        if (e != hipSuccess) fail("scratch for the packed row nibbles", e);
    }
    if (rc == 0 && dev_pack_desc(d_udesc, d_urow, d_ucol, (const int4 *)d_map, (int)pair_map.size(), d_packed, d_prow, const_cast<uint4 *>(S.ucol)) != 0) rc = -3;
    if (rc == 0 && pooled && !wide && K.desc_dict != 0 && NUP > 0) {   // 8-B descriptors + pattern dictionary (the host builder's rule)
        std::vector<uint4> pats;
        bool over = false;
        if (dev_pool_dict(d_packed, d_prow, NUP, (size_t)1 << DICT_MAX_BITS, pats, &over) != 0) rc = -3;
        else if (!over) {
            rc |= plan->upload(pats.data(), pats.size(), &S.pdict);
            const int bb = pool_word_base_bits((int)pats.size());   // (4-byte words under the host builder's rule)
            const bool word = K.desc_dict != 2 && POOL_STRIP_ROWS <= 4 && bb >= 4 && (long long)T->tilen * 16 <= (1ll << bb);
            if (word) rc |= plan->reserve((size_t)NUP, reinterpret_cast<const unsigned **>(&S.udesc));
            else rc |= plan->reserve((size_t)NUP, reinterpret_cast<const uint2 **>(&S.udesc));
            if (rc == 0 && dev_pool_compact(d_packed, d_prow, NUP, S.pdict, (int)pats.size(), word ? bb : 0, const_cast<UDesc *>(S.udesc)) != 0) rc = -3;
            pool_dict = true;
            if (word) S.cb_bits = bb;
        }
    }
    if (rc == 0 && pooled && !wide && !pool_dict) {
        rc |= plan->reserve((size_t)NUP, &S.urow);
        if (rc == 0 && NUP > 0 && (e = hipMemcpy(const_cast<URow *>(S.urow), d_prow, (size_t)NUP * sizeof(URow), hipMemcpyDeviceToDevice)) != hipSuccess) fail("row nibble copy", e);
    }
    // 4-B descriptors + pattern dictionary under the host builder's conditions (below); the distinct patterns come from a sort + run-length encoding of the packed descriptors
    const bool dict_pays = K.desc_dict > 0 ? true : 8LL * NUP * 50 >= NUP * (12 + 16LL * sv) + NC * (sv + 4LL);
    if (rc == 0 && K.desc_dict != 0 && dict_pays && !pooled && NUP > 0) {
        const int cb_bits = std::max(1, 32 - __builtin_clz((unsigned)std::max(1, T->tilen - 1)));
        const int pid_bits = std::min(DICT_MAX_BITS, 27 - cb_bits);
        if (pid_bits >= 1) {
            std::vector<uint4> dict;   // (the host builder's order: shift code, then nibbles)
            DictRanges ranges;
            bool over = false;
            if (dev_dict_patterns(d_packed, NUP, (size_t)1 << pid_bits, dict, &ranges, &over) != 0) rc = -3;
            else if (!over) {
                rc |= plan->upload(dict.data(), dict.size(), &S.udict);
                rc |= plan->reserve((size_t)NUP, reinterpret_cast<const unsigned **>(&S.udesc));
                if (rc == 0 && dev_compact_desc(d_packed, NUP, S.udict, ranges, cb_bits, reinterpret_cast<unsigned *>(const_cast<UDesc *>(S.udesc))) != 0) rc = -3;
                S.cb_bits = cb_bits;
            }
        }
    }
    if (rc == 0 && S.cb_bits == 0 && !pool_dict) {
        rc |= plan->reserve((size_t)NUP, &S.udesc);
        if (rc == 0 && NUP > 0 && (e = hipMemcpy(const_cast<UDesc *>(S.udesc), d_packed, (size_t)NUP * sizeof(UDesc), hipMemcpyDeviceToDevice)) != hipSuccess) fail("descriptor copy", e);
    }
    plan->info[TILESPMV_INFO_DESC_BYTES] = desc_bytes();
    if (rc == 0 && !pooled) { unsigned long long hist[8]; if (dev_shift_histogram(d_packed, NUP, hist) != 0) rc = -3; else plan->info[TILESPMV_INFO_DERIVED_UNITS] = (long long)hist[UNIT_DERIVED_CODE]; }
    // the value pass (as in host mode: k_pair_values), its source already on the device
    rc |= plan->reserve((size_t)NUP * 16, &S.uval);
    if (rc == 0) {
        e = launch_pair_values(d_uval, const_cast<val_t *>(S.uval), (const int4 *)d_map, (int)pair_map.size());
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) fail("value pass", e);
    }
    for (void *q : {(void *)d_map, (void *)d_packed, (void *)d_prow, (void *)d_udesc, (void *)d_urow, (void *)d_uval, (void *)d_ucol}) if (q) (void)hipFree(q);
    d_udesc = nullptr; d_urow = nullptr; d_uval = nullptr; d_ucol = nullptr;
    S.udesc_cb = S.udesc;
    plan->info[TILESPMV_INFO_UPLOAD_US] = up0 + (long long)(now_us() - t0);
}

void StreamBuilder::encode()
{
    const unsigned long long d0 = plan->digest;
    // ---- final HBM form of the unit streams.  Descriptors: 12 B (the duplicate of word 0 is dropped).  Values: the
    // units of one task are stored in GROUPS of G = 16 / sizeof(value) units (2 in fp64, 4 in fp32) — the values of
    // the G units interleaved per row, so that a lane fetches G units with one 16-byte load (row r of the group at
    // +16 r bytes).  A task whose unit count is not a multiple of G gets padding units (zero values, never executed:
    // unit_end excludes them) so that its last group exists.
    constexpr long long G = UNIT_GROUP;
    auto padded = [&](long long n) { return (n + G - 1) / G * G; };
    NUP = 0;
    for (const STask &k : tasks) NUP += padded(k.unit_end - k.unit_begin);
    if (NUP > INT32_MAX) {
        fprintf(stderr, "tilespmv: shard too large for 32-bit unit ids\n");
        rc = -2;
        return;
    }
    if (DT) { encode_device(); return; }
    {
        std::vector<UDesc> packed((size_t)NUP, UDesc{0u, 0u, 0u});
        std::vector<URow> packed_row(pooled && !wide ? (size_t)NUP : 0, URow{0u, 0u});
        std::vector<uint4> packed_col(wide ? (size_t)NUP : 0, make_uint4(0u, 0u, 0u, 0u));
        // The value pass (the plan's largest array, permuted into groups per task) runs on the DEVICE unless this is a layout-digest build or TILESPMV_ENCODE_ON_HOST=1 asks for
        // the host pass — which stays as the checker: TILESPMV_ENCODE_CHECK=1 runs both and compares the device's stream with the host's, byte for byte (tests/test_gpu_parity.py)
        const bool encode_check = !plan->dry && env_int("TILESPMV_ENCODE_CHECK", 0) != 0;
        const bool on_device = !plan->dry && NUP > 0 && env_int("TILESPMV_ENCODE_ON_HOST", 0) == 0;
        val_t *paired = (on_device && !encode_check) ? nullptr : zalloc<val_t>((size_t)NUP * 16);
        std::vector<int4> pair_map(on_device ? tasks.size() : 0);
        std::vector<long long> new_begin(tasks.size());
        old_begin.assign(tasks.size(), 0);
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
                    if (pooled && !wide) packed_row[(size_t)(nb + j)] = URow{h_urow[(size_t)(ub + j)].x, h_urow[(size_t)(ub + j)].y};
                    if (wide) packed_col[(size_t)(nb + j)] = h_ucol[(size_t)(ub + j)];
                    if (paired) {
                        const val_t *src = h_uval + (ub + j) * 16;
                        val_t *dst = paired + (nb + j / G * G) * 16 + (j % G);
                        for (int r = 0; r < 16; r++) dst[G * r] = src[r];
                    }
                }
                if (on_device) pair_map[(size_t)i] = make_int4((int)ub, (int)nb, (int)n, 0);
                if (n > 0) { k.unit_begin = (int)nb; k.unit_end = (int)(nb + n); }
            }
        });
        if (pooled && !wide && getenv("TILESPMV_POOL_PATTERN_STAT")) {   // (study: how many distinct (column nibbles, row nibbles) patterns do the pooled units of this shard use?)
            std::unordered_map<std::string, long long> cnt;
            for (long long u = 0; u < NUP; u++) { unsigned w[4] = {packed[(size_t)u].n0, packed[(size_t)u].n1, packed_row[(size_t)u].r0, packed_row[(size_t)u].r1}; cnt[std::string((const char *)w, 16)]++; }
            std::vector<long long> c; for (auto &kv : cnt) c.push_back(kv.second);
            std::sort(c.begin(), c.end(), std::greater<long long>());
            long long top1k = 0, top4k = 0, top64k = 0; for (size_t i = 0; i < c.size(); i++) { if (i < 1024) top1k += c[i]; if (i < 4096) top4k += c[i]; if (i < 65536) top64k += c[i]; }
            fprintf(stderr, "tilespmv: pooled units: %lld units, %zu distinct 16-byte patterns; the 1,024 / 4,096 / 65,536 most frequent cover %.1f / %.1f / %.1f %% of the units\n", NUP, c.size(), 100.0 * top1k / std::max(1LL, NUP), 100.0 * top4k / std::max(1LL, NUP), 100.0 * top64k / std::max(1LL, NUP));
        }
        // ---- 4-B descriptors where the units of the shard use few distinct column patterns (stencil-like shards: 4 patterns in the
        // 5- and 7-point grids, 36 in the KKT stand-in): column block | pattern id << cb_bits | flags << 27, the patterns (the
        // two nibble words) in a dictionary the kernels gather from.  Not for x-window plans (their descriptors hold slots).
        S.udict = nullptr; S.cb_bits = 0;
        std::vector<uint4> dict;   // (nibbles of rows 0-7, of rows 8-15, window shift << UNIT_SHIFT_SHIFT, 0), ascending (shift code, nibbles)
        std::vector<unsigned> compact;
        // ... and only where it pays: 8 bytes per unit must be at least 2 % of the streams (an entry-dominated plan with a handful of units would only buy the dictionary
        // hop at the start of every strip: webbase-1M stand-in 13.2 -> 13.6 us); desc_dict = 1 asks for it wherever it is possible
        const bool dict_pays = K.desc_dict > 0 ? true : 8LL * NUP * 50 >= NUP * (12 + 16LL * sv) + NC * (sv + 4LL);
        if (K.desc_dict != 0 && dict_pays && !pooled && NUP > 0) {
            const int cb_bits = std::max(1, 32 - __builtin_clz((unsigned)std::max(1, T->tilen - 1)));
            const int pid_bits = std::min(DICT_MAX_BITS, 27 - cb_bits);
            if (pid_bits >= 1) {
                const size_t cap = (size_t)1 << pid_bits;
                // a pattern = the unit's window shift (units that took list entries: plan_tile_ops.h; code = word 0 >> UNIT_SHIFT_SHIFT) and its 16 column nibbles
                typedef std::pair<unsigned, unsigned long long> Pat;
                std::vector<std::array<std::unordered_set<unsigned long long>, 8>> local((size_t)host_threads());
                std::atomic<int> over(0);
                parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int th) {
                    if (over.load(std::memory_order_relaxed)) return;
                    auto &L = local[(size_t)th];
                    size_t have = 0;
                    for (auto &q : L) have += q.size();
                    for (int64_t u = b; u < e; u++) {
                        if (L[packed[(size_t)u].w0 >> UNIT_SHIFT_SHIFT].insert(((unsigned long long)packed[(size_t)u].n0 << 32) | packed[(size_t)u].n1).second) have++;
                        if (have > cap) { over.store(1); return; }
                    }
                });
                std::vector<Pat> all;
                if (!over.load()) {
                    for (auto &L : local) for (unsigned sc = 0; sc < 8; sc++) for (unsigned long long k : L[sc]) all.push_back(Pat(sc, k));
                    std::sort(all.begin(), all.end());
                    all.erase(std::unique(all.begin(), all.end()), all.end());
                }
                if (!over.load() && all.size() <= cap) {
                    dict.resize(all.size());
                    for (size_t i = 0; i < all.size(); i++) dict[i] = make_uint4((unsigned)(all[i].second >> 32), (unsigned)(all[i].second & 0xffffffffull), all[i].first << UNIT_SHIFT_SHIFT, 0u);
                    compact.resize((size_t)NUP);
                    const unsigned cbmask = (1u << cb_bits) - 1u;
                    parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int) {
                        for (int64_t u = b; u < e; u++) {
                            const UDesc &d = packed[(size_t)u];
                            const Pat key(d.w0 >> UNIT_SHIFT_SHIFT, ((unsigned long long)d.n0 << 32) | d.n1);
                            const unsigned pid = (unsigned)(std::lower_bound(all.begin(), all.end(), key) - all.begin());
                            compact[(size_t)u] = (d.w0 & cbmask) | (pid << cb_bits) | ((d.w0 >> UNIT_FLAG_SHIFT) << 27);
                        }
                    });
                    S.cb_bits = cb_bits;
                }
            }
        }
        // ---- pooled plans: 8-B descriptors (word 0, pattern id) + a dictionary of 16-byte patterns (the unit's 16 column nibbles and 16 row nibbles) where the shard's units use at most
        // 2^DICT_MAX_BITS distinct ones — natural-order meshes use a few dozen (27-point hex mesh x 3 unknowns: 54; tetrahedral: 31), window-shuffled ones a hundred thousand and keep
        // the 20-byte form.  12 of 148 bytes per unit: always worth it where it applies (desc_dict = 0 switches it off).  Patterns in ascending (n0, n1, r0, r1) order.
        S.pdict = nullptr; pool_dict = false;
        std::vector<uint4> pdict;
        std::vector<uint2> compact2;
        std::vector<unsigned> compact1;
        if (pooled && !wide && K.desc_dict != 0 && NUP > 0) {
            typedef std::array<unsigned, 4> Pat;
            const size_t cap = (size_t)1 << DICT_MAX_BITS;
            std::vector<std::set<Pat>> local((size_t)host_threads());
            std::atomic<int> over(0);
            parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int th) {
                if (over.load(std::memory_order_relaxed)) return;
                std::set<Pat> &L = local[(size_t)th];
                Pat last{{~0u, ~0u, ~0u, ~0u}};
                for (int64_t u = b; u < e; u++) {
                    const Pat q{{packed[(size_t)u].n0, packed[(size_t)u].n1, packed_row[(size_t)u].r0, packed_row[(size_t)u].r1}};
                    if (q == last) continue;
                    last = q;
                    L.insert(q);
                    if (L.size() > cap) { over.store(1); return; }
                }
            });
            std::vector<Pat> all;
            if (!over.load()) {
                for (auto &L : local) all.insert(all.end(), L.begin(), L.end());
                std::sort(all.begin(), all.end());
                all.erase(std::unique(all.begin(), all.end()), all.end());
            }
            if (!over.load() && all.size() <= cap) {
                pdict.resize(all.size());
                for (size_t i = 0; i < all.size(); i++) pdict[i] = make_uint4(all[i][0], all[i][1], all[i][2], all[i][3]);
                compact2.resize((size_t)NUP);
                parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int) {
                    for (int64_t u = b; u < e; u++) {
                        const Pat q{{packed[(size_t)u].n0, packed[(size_t)u].n1, packed_row[(size_t)u].r0, packed_row[(size_t)u].r1}};
                        compact2[(size_t)u] = make_uint2(packed[(size_t)u].w0, (unsigned)(std::lower_bound(all.begin(), all.end(), q) - all.begin()));
                    }
                });
                pool_dict = true;
                // round 6: one 4-byte word per unit where window base, pattern id and tile-row fit it (hip_plan.h; desc_dict = 2 keeps the pairs)
                const int bb = pool_word_base_bits((int)all.size());
                if (K.desc_dict != 2 && POOL_STRIP_ROWS <= 4 && bb >= 4 && (long long)T->tilen * 16 <= (1ll << bb)) {
                    compact1.resize((size_t)NUP);
                    parallel_chunks((int64_t)NUP, 1 << 16, [&](int64_t b, int64_t e, int) {
                        for (int64_t u = b; u < e; u++) {
                            const uint2 c = compact2[(size_t)u];
                            compact1[(size_t)u] = (c.x & POOL_BASE_MASK) | (c.y << bb) | ((c.x >> POOL_KR_SHIFT) << POOL_WORD_KR_SHIFT);
                        }
                    });
                    S.cb_bits = bb;
                }
            }
        }
        S.urow = nullptr; S.ucol = nullptr; S.pooled = pooled ? 1 : 0;
        if (pool_dict) {
            if (S.cb_bits > 0) rc |= plan->upload(compact1.data(), compact1.size(), reinterpret_cast<const unsigned **>(&S.udesc));
            else rc |= plan->upload(compact2.data(), compact2.size(), reinterpret_cast<const uint2 **>(&S.udesc));
            rc |= plan->upload(pdict.data(), pdict.size(), &S.pdict);
        } else if (S.cb_bits > 0) {
            rc |= plan->upload(compact.data(), compact.size(), reinterpret_cast<const unsigned **>(&S.udesc));
            rc |= plan->upload(dict.data(), dict.size(), &S.udict);
        } else rc |= plan->upload(packed.data(), (size_t)NUP, &S.udesc);
        plan->info[TILESPMV_INFO_DESC_BYTES] = desc_bytes();
        if (!pooled) { long long nder = 0; for (long long u = 0; u < NUP; u++) nder += (packed[(size_t)u].w0 >> UNIT_SHIFT_SHIFT) == UNIT_DERIVED_CODE; plan->info[TILESPMV_INFO_DERIVED_UNITS] = nder; }
        if (pooled && !wide && !pool_dict) rc |= plan->upload(packed_row.data(), (size_t)NUP, &S.urow);
        if (wide) rc |= plan->upload(packed_col.data(), (size_t)NUP, &S.ucol);
        if (on_device) {
            // emitted values as they are -> a scratch buffer on the device; one workgroup per task writes them to their final place in the plan's arena
            const double t0 = now_us();
            void *d_src = nullptr, *d_map = nullptr;
            const size_t src_b = (size_t)NU * 16 * sizeof(val_t), map_b = pair_map.size() * sizeof(int4);
            rc |= plan->reserve((size_t)NUP * 16, &S.uval);
            hipError_t e = hipMalloc(&d_src, std::max<size_t>(src_b, 16));
            if (e == hipSuccess) e = hipMalloc(&d_map, std::max<size_t>(map_b, 16));
            if (e == hipSuccess) e = hipMemcpy(d_src, h_uval, src_b, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_map, pair_map.data(), map_b, hipMemcpyHostToDevice);
            if (e == hipSuccess && rc == 0) e = launch_pair_values((const val_t *)d_src, const_cast<val_t *>(S.uval), (const int4 *)d_map, (int)pair_map.size());
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess && encode_check) {
                std::vector<val_t> back((size_t)NUP * 16);
                e = hipMemcpy(back.data(), S.uval, back.size() * sizeof(val_t), hipMemcpyDeviceToHost);
                if (e == hipSuccess && memcmp(back.data(), paired, back.size() * sizeof(val_t)) != 0) { fprintf(stderr, "tilespmv: internal error: the device's value stream differs from the host's\n"); rc = -6; }
                else if (e == hipSuccess && getenv("TILESPMV_PLAN_VERBOSE")) fprintf(stderr, "tilespmv: encode check: %lld units, device value stream == host value stream\n", NUP);
            }
            if (d_src) (void)hipFree(d_src);
            if (d_map) (void)hipFree(d_map);
            if (e != hipSuccess) { fprintf(stderr, "tilespmv: value pass on the device failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); rc = -3; }
            plan->info[TILESPMV_INFO_UPLOAD_US] += (long long)(now_us() - t0);
        } else rc |= plan->upload(paired, (size_t)NUP * 16, &S.uval);
        free_later({paired, h_uval}, (size_t)NUP * 16 * sizeof(val_t));   // (h_uval was read for the last time above)
        h_uval = nullptr;
        S.udesc_cb = S.udesc;
    }
    if (hashing()) { Hash h; h.num((long long)(plan->digest ^ d0)); h.num(NUP); h.num(S.cb_bits); h.num(S.pooled); if (pool_dict) h.num(8); stage_done(TILESPMV_STAGE_ENCODE, h); }   // (everything this stage produces is uploaded: the running upload digest covers it)
}

void StreamBuilder::entries()
{
    const unsigned long long d0 = plan->digest;
    S.wg_coo = nullptr; S.grec = nullptr; S.gbase = nullptr; S.dest_bits = 11;
    n_rec = 0; n_chunk = 0; n_groups = 0;
    if (entry_mode != 0) {
        const size_t GS = entry_mode == 2 ? (size_t)wg_strips : 4;   // tasks whose lists are merged: one workgroup's or one wavefront's
        const int slab_shift = pooled ? (POOL_STRIP_ROWS > 4 ? 7 : 6) : 7;   // a strip's slab of s_y: POOL_STRIP_ROWS x 16 values in pooled plans, STRIP_MAX_ROWS x 16 otherwise
        const int dest_bits = entry_mode == 2 ? (wg_strips == 32 ? 12 : 4 + slab_shift) : 9;   // strip-in-group | row-in-strip | row (4)
        S.dest_bits = dest_bits;
        const size_t nwg = (tasks.size() + GS - 1) / GS;
        const size_t NP = (size_t)x_panels;          // column panels: a group's list stays ONE column-ordered list; panel p is the run [panel_off[p], panel_off[p + 1]) of it
        std::vector<std::vector<ERec>> grp_rec(nwg);
        std::vector<std::vector<unsigned>> grp_base(nwg);
        std::vector<int> h_panel_off(NP > 1 ? nwg * (NP + 1) : 0, 0);   // (relative to the list's begin here; absolute record indices below)
        std::atomic<int> bad(0);
        std::atomic<long long> scattered(0);
        DevLists dlists;   // device mode: the lists are merged, ordered and packed on the device (hip_plan_device.h: one stable sort by (group, column), then the same packing function)
        if (DT) {
            if (dev_entry_lists(S.cval, S.ccol, S.crow, NC, tasks, (int)GS, slab_shift, dest_bits, entry_mode == 2, x_panels, panel_shift, &dlists) != 0) { rc = -3; return; }
            scattered.store(dlists.scattered);
            if (NP > 1) h_panel_off = dlists.panel_off;
        }
        parallel_chunks(DT ? 0 : (int64_t)nwg, 64, [&](int64_t b, int64_t e, int) {
            std::vector<std::pair<unsigned long long, unsigned>> key;   // (column << 32 | position in strip / list order, destination)
            std::vector<int> src;
            std::vector<PEnt> ents;
            for (int64_t w = b; w < e; w++) {
                key.clear(); src.clear();
                long long own_lo = LLONG_MAX, own_hi = LLONG_MIN;   // columns "around the group's own rows": [16 first tile-row, 16 (last tile-row + 1))
                for (size_t t = GS * (size_t)w; t < std::min(tasks.size(), GS * (size_t)w + GS); t++) {
                    own_lo = std::min<long long>(own_lo, 16LL * tasks[t].row); own_hi = std::max<long long>(own_hi, 16LL * (tasks[t].row + std::max(1, tasks[t].nrows)));
                    for (int q = tasks[t].coo_begin; q < tasks[t].coo_end; q++) {   // column-major order; ties keep strip / list order
                        key.push_back({((unsigned long long)(unsigned)h_ccol[(size_t)q] << 32) | (unsigned long long)key.size(),
                                       (unsigned)((t & (GS - 1)) << slab_shift) | (unsigned)h_crow[(size_t)q]});
                        src.push_back(q);
                    }
                }
                std::sort(key.begin(), key.end());
                ents.resize(key.size());
                for (size_t i = 0; i < key.size(); i++) {
                    const int q = src[(size_t)(key[i].first & 0xFFFFFFFFull)];
                    ents[i] = PEnt{(unsigned)h_ccol[(size_t)q], key[i].second, h_cval[q]};
                }
                if (entry_mode == 2) {   // plan fact: entries far from the group's own rows (scattered gathers)
                    long long far = 0;
                    for (const PEnt &en : ents) far += !((long long)en.col >= own_lo - 2048 && (long long)en.col < own_hi + 2048);
                    scattered.fetch_add(far, std::memory_order_relaxed);
                }
                if (!pack_list(ents, dest_bits, grp_rec[(size_t)w], grp_base[(size_t)w], plan->dry)) bad++;
                if (NP > 1) {
                    // where each panel begins in the PACKED list: records are in column order except that the null padding of a chunk closed early repeats the chunk's first
                    // column — padding counts as part of the panel of the record before it (it adds 0 * x[.] to the group's first row whichever pass executes it)
                    panel_offsets(grp_rec[(size_t)w].data(), (long long)grp_rec[(size_t)w].size(), grp_base[(size_t)w].data(), dest_bits, panel_shift, (int)NP, &h_panel_off[(size_t)w * (NP + 1)]);
                }
            }
        });
        if (bad.load()) { fprintf(stderr, "tilespmv: internal error: %d packed entry lists do not decode to their entries\n", bad.load()); rc = -6; }
        std::vector<int4> wg((size_t)nwg);
        if (DT) { wg = dlists.wg; n_rec = dlists.n_rec; n_chunk = dlists.n_chunk; }
        else
        for (size_t w = 0; w < nwg; w++) {
            wg[w] = make_int4((int)n_rec, (int)(n_rec + (long long)grp_rec[w].size()), (int)n_chunk, 0);
            if (NP > 1) for (size_t q = 0; q <= NP; q++) h_panel_off[w * (NP + 1) + q] += (int)n_rec;
            n_rec += (long long)grp_rec[w].size(); n_chunk += (long long)grp_base[w].size();
        }
        if (n_rec > INT32_MAX) { fprintf(stderr, "tilespmv: shard too large for 32-bit entry ids\n"); rc = -2; n_rec = 0; }
        std::vector<ERec> g_rec(DT ? 0 : (size_t)n_rec);
        std::vector<unsigned> g_base(DT ? 0 : (size_t)n_chunk);
        if (rc == 0 && !DT)
            parallel_chunks((int64_t)nwg, 256, [&](int64_t b, int64_t e, int) {
                for (int64_t w = b; w < e; w++) {
                    if (!grp_rec[(size_t)w].empty()) memcpy(&g_rec[(size_t)wg[(size_t)w].x], grp_rec[(size_t)w].data(), grp_rec[(size_t)w].size() * sizeof(ERec));
                    if (!grp_base[(size_t)w].empty()) memcpy(&g_base[(size_t)wg[(size_t)w].z], grp_base[(size_t)w].data(), grp_base[(size_t)w].size() * sizeof(unsigned));
                }
            });
        n_groups = (long long)nwg;
        plan->info[TILESPMV_INFO_SCATTERED_ENTRIES] = scattered.load();
        // rows whose sums ONE panel pass adds to y (16 bytes per row, read + written): the strips with entries of the groups that have any record in that panel; averaged over the panels
        panel_rmw_rows = 0;
        if (NP > 1) {
            long long acc_rows = 0;
            for (size_t w = 0; w < nwg; w++) {
                long long rows_w = 0;
                for (size_t t = GS * w; t < std::min(tasks.size(), GS * w + GS); t++)
                    if (tasks[t].coo_end > tasks[t].coo_begin) rows_w += 16LL * std::max(1, tasks[t].nrows);
                for (size_t q = 1; q < NP; q++)
                    if (h_panel_off[w * (NP + 1) + q + 1] > h_panel_off[w * (NP + 1) + q]) acc_rows += rows_w;
            }
            panel_rmw_rows = acc_rows;    // at the finest panels; a launch that merges m panels per pass touches about 1 / m of it (tilespmv_plan_info reports the model of the form chosen)
        }
        rc |= plan->upload(wg.data(), wg.size(), &S.wg_coo);
        if (DT) {   // records and bases are on the device already: into the plan's arena
            rc |= plan->reserve((size_t)n_rec, &S.grec);
            rc |= plan->reserve((size_t)n_chunk, &S.gbase);
            hipError_t e = hipSuccess;
            if (rc == 0 && n_rec > 0) e = hipMemcpy(const_cast<ERec *>(S.grec), dlists.d_rec, (size_t)n_rec * sizeof(ERec), hipMemcpyDeviceToDevice);
            if (rc == 0 && e == hipSuccess && n_chunk > 0) e = hipMemcpy(const_cast<unsigned *>(S.gbase), dlists.d_base, (size_t)n_chunk * sizeof(unsigned), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { fprintf(stderr, "tilespmv: device plan build: entry lists into the plan: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); rc = -3; }
            dlists.release();
        } else {
            rc |= plan->upload(g_rec.data(), g_rec.size(), &S.grec);
            rc |= plan->upload(g_base.data(), g_base.size(), &S.gbase);
        }
        S.panel_off = nullptr;
        if (NP > 1) rc |= plan->upload(h_panel_off.data(), h_panel_off.size(), &S.panel_off);
    }
    S.x_panels = x_panels; S.n_groups = (int)n_groups;
    S.panel_merge = (x_panels > 1 && K.x_panel_merge > 0) ? std::min(K.x_panel_merge, x_panels) : 0;
    // column slices pinned to XCDs (DevStream::slice_passes): same lists and offsets, another launch form; never when the caller asked for reproducible sums
    const bool slices_allowed = x_panels > 1 && K.entry_ordered != 1;
    S.slice_passes = (slices_allowed && K.x_slice_passes > 0) ? std::min(K.x_slice_passes, 8) : 0;
    plan->list_records = n_rec;
    S.slice_ct = slice_trip_records(n_rec, (int)n_groups, std::max(1, S.slice_passes));
    if (S.slice_passes > 0) S.panel_merge = 0;
    plan->panel_calibrate = x_panels > 1 && K.x_panel_merge < 0 && S.slice_passes == 0;
    plan->slice_calibrate = slices_allowed && K.x_slice_passes < 0 && K.x_panel_merge < 0;   // (a caller who fixes the panels per pass has chosen the form)
    plan->panel_rmw_rows = panel_rmw_rows;
    plan->info[TILESPMV_INFO_X_PANELS] = S.panel_merge > 0 ? (x_panels + S.panel_merge - 1) / S.panel_merge : 1;   // launches of the entry part
    if (hashing()) { Hash h; h.num((long long)(plan->digest ^ d0)); h.num(n_rec); h.num(n_chunk); h.num(n_groups); h.num(S.dest_bits); h.num(x_panels); stage_done(TILESPMV_STAGE_ENTRIES, h); }
}

void StreamBuilder::finish(long long &n_tasks, long long &model_bytes)
{
    if (!DT) {   // (device mode: EMIT wrote them into the arena)
        rc |= plan->upload(h_cval, (size_t)NC, &S.cval);
        rc |= plan->upload(h_ccol.data(), (size_t)NC, &S.ccol);
        rc |= plan->upload(h_crow.data(), (size_t)NC, &S.crow);
    }
    DevPlan &D = plan->dev;  // heavy tiles reuse the first-generation streams + kernel (accumulate mode)
    rc |= plan->upload(h_hdesc.data(), (size_t)NH, &D.desc);
    rc |= plan->upload(h_hval, (size_t)NHV, &D.val);
    rc |= plan->upload(h_hidx, (size_t)NHI, &D.idx);
    rc |= plan->upload(htasks.data(), htasks.size(), &D.task);
    D.ntasks = (int)htasks.size();
    rc |= plan->upload(tasks.data(), tasks.size(), &S.task);
    if (!DT) {
        rc |= plan->upload(h_dcb.data(), (size_t)ND, &plan->dn.cb);
        rc |= plan->upload(h_dval, (size_t)ND * 256, &plan->dn.val);
    }
    rc |= plan->upload(drows.data(), drows.size(), &plan->dn.rows);
    plan->dn.nrows = (int)drows.size();
    for (size_t i = 0; i < drows.size(); i++) {   // k_dense_mfma walks consecutive records as one flat run of tiles: their ranges must be non-empty and follow each other
        const DenseRow &dr = drows[i];
        if (dr.tile_end <= dr.tile_begin || (i > 0 && dr.tile_begin != drows[i - 1].tile_end)) { fprintf(stderr, "tilespmv: internal error: dense row records are not one contiguous run of tiles\n"); rc = -6; break; }
    }
    release();
    S.ntasks = (int)tasks.size();
    // small grids: XCD windows of 8 x 8 workgroups — with the default 8 x 32 a grid of under 256 workgroups has no full window at all and is dealt round-robin, so neighbouring strips
    // never share an XCD's L2 (scircuit stand-in, 246 workgroups: 6.95 -> 6.55 us; webbase stand-in, 750: 13.05 -> 12.75; profiles/r05_small_grid_forms.txt)
    if (!K.xcd_from_caller && ((long long)tasks.size() + wg_strips - 1) / wg_strips < 1024) plan->xcd_chunk = 8;
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
        const long long stream_b = NU * (desc_bytes() + 16LL * sv) + NC * (sv + 4LL), y_b = 16LL * ntr * sv;
        S.y_streaming = K.y_store >= 0 ? (K.y_store != 0) : (y_b * 20 >= stream_b);   // >= 5 %
    }
    plan->info[TILESPMV_INFO_ENTRY_MODE] = entry_mode;
    plan->info[TILESPMV_INFO_ENTRY_ORDERED] = (entry_mode != 2 || coo_ordered) ? 1 : 0;
    plan->info[TILESPMV_INFO_STRIP_COST] = target;
    plan->info[TILESPMV_INFO_WG_STRIPS] = wg_strips;
    plan->info[TILESPMV_INFO_CSR_FORM] = csr_form;
    plan->pooled = pooled;
    {   // entry slab of the multi-vector kernel: shards with >= 3 entries per tile-row (strips then regularly hold more than the 16 entries that travel with the prologue)
        int used = 1;
        for (const STask &k : tasks) used = std::max(used, k.nrows);
        const int slab_env = env_int("TILESPMV_MV_SLAB", -1);   // (experiment knob: 0 off, 1 on wherever entries exist)
        plan->mv_slab_rows = (slab_env == 0 || NC == 0) ? 0 : (slab_env > 0 || NC >= 3LL * ntr) ? used : 0;
    }
    plan->mv_by_columns = entry_dominated && target >= 800;   // (small strips hold few entries each: scircuit-like 18 / 22 / 32 us native against 22 / 41 / 78 us)
    n_tasks = (long long)tasks.size();
    model_bytes = NUP * (desc_bytes() + 16LL * sv) + (entry_mode == 0 ? NC * (sv + 5LL) : n_rec * (long long)sizeof(ERec) + n_chunk * 4 + n_groups * 16) + NH * 8 + NHV * sv + NHI + n_tasks * (long long)sizeof(STask) +
                  (long long)htasks.size() * ((long long)sizeof(Task) + 32LL * sv) +  // whole-tile passes re-read and re-write their rows of y
                  ND * (4 + 256LL * sv) + (long long)drows.size() * (16 + 32LL * sv);
    // The once-read streams (values, entry records) are loaded nontemporally when the launch moves clearly more than the Infinity Cache holds: they then do
    // not displace x in the L2s / the Infinity Cache — config 4 0.182 -> 0.164-0.166 ms, 7-pt 256^3 0.231 -> 0.211, KKT fp32 0.253 -> 0.236-0.243, power-law
    // 8 M 0.103 -> 0.095 — while a plan that (nearly) fits keeps the default policy, because its streams come back from the Infinity Cache on the next SpMV:
    // nontemporal loses 2 % at 340 MB (5-pt 2400^2), 15 % at 180-300 MB (power-law 3-5 M rows), 6-8 % on webbase-1M; it wins from 500 MB up (5-pt 2896^2 +4 %,
    // power-law 8 M +8 %, 5-pt 3400^2 +10 %).  Descriptors, tasks and per-strip entry lists stay on the default policy (nontemporal: config 4 0.164 -> 0.170-0.173).
    // profiles/r03_nontemporal_streams.txt.  Entry mode 1 = small grids.
    {
        const long long launch_b = model_bytes + ((long long)colA + 16LL * ntr) * sv;
        S.nt_stream = (entry_mode != 1 && (K.nt_stream >= 0 ? K.nt_stream != 0 : launch_b > NT_STREAM_MIN_BYTES)) ? 1 : 0;
    }
    plan->info[TILESPMV_INFO_NT_STREAM] = S.nt_stream;
    if (hashing()) {
        Hash h;
        for (long long v : {model_bytes, n_tasks, (long long)S.y_streaming, (long long)S.nt_stream, (long long)S.coo_ordered, (long long)plan->mv_slab_rows, (long long)plan->mv_by_columns, (long long)S.coo_heavy_min}) h.num(v);
        stage_done(TILESPMV_STAGE_FINISH, h);
    }
}

}  // namespace

int tilespmv::build_stream(tilespmv_plan *plan, const Knobs &K, const Tile_matrix *T, int rowA, int colA, int tr0, int tr1, bool coo_in_tile,
                           bool dense_mfma, const std::vector<long long> &hyb_off,
                           std::vector<FixRow> &fix, int &npartial, long long &n_tasks, long long &model_bytes, const DevTile *DT)
{
    StreamBuilder B(plan, K, T, rowA, colA, tr0, tr1, coo_in_tile, dense_mfma, hyb_off, fix, npartial, DT);
    const bool verbose = getenv("TILESPMV_PLAN_VERBOSE") != nullptr;
    double t_prev = now_us(), up_prev = (double)plan->info[TILESPMV_INFO_UPLOAD_US];
    std::string times;
    auto lap = [&](const char *name) {   // host milliseconds of the stage (its uploads counted apart)
        if (!verbose) return;
        const double t = now_us(), up = (double)plan->info[TILESPMV_INFO_UPLOAD_US];
        char buf[96];
        snprintf(buf, sizeof(buf), " %s %.1f (+%.1f upload)", name, (t - t_prev - (up - up_prev)) * 1e-3, (up - up_prev) * 1e-3);
        times += buf; t_prev = t; up_prev = up;
    };
    B.count(); lap("count");
    if (B.rc) return B.rc;
    B.choose(); lap("choose");
    B.cut(); lap("cut");
    B.emit(); lap("emit");
    if (B.rc == -3) return B.rc;   // (device mode: a HIP error is final)
    B.order(); lap("order");
    B.encode(); lap("encode");
    if (B.rc == -2 || B.rc == -3) return B.rc;   // (shard too large for 32-bit unit ids)
    B.entries(); lap("entries");
    B.finish(n_tasks, model_bytes); lap("finish");
    if (verbose) fprintf(stderr, "tilespmv: unit-stream layout, ms per stage:%s\n", times.c_str());
    return B.rc;
}
