/*
 * tilespmv.h — C ABI of the MI355X-native TileSpMV engine (libtilespmv_f64.so / libtilespmv_f32.so).
 *
 * Drop-in boundary for the y = A*x hot path of SuperScientificSoftwareLaboratory/TileSpMV.
 * The reference has no FFI layer: "the API" is the set of C functions its driver calls
 * (reference src/main.cu:63,87,142,165) plus the CLI.  Every entry point below names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Value type is a build-time choice exactly like the reference (src/common.h:12-14,
 * src/Makefile:5,23):  libtilespmv_f64.so  <=>  -D MAT_VAL_TYPE=double
 *                      libtilespmv_f32.so  <=>  -D MAT_VAL_TYPE=float
 * Both libraries export the same symbol names.
 */
#ifndef TILESPMV_H_
#define TILESPMV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MAT_VAL_TYPE
#define MAT_VAL_TYPE double /* reference src/common.h:12-14 */
#endif
#ifndef MAT_PTR_TYPE
#define MAT_PTR_TYPE int /* reference src/common.h:25-27 */
#endif

/* Compile-time constants the tile format is defined by (reference src/common.h:37-63). */
#define TILESPMV_BLOCK_SIZE 16       /* BLOCK_SIZE */
#define TILESPMV_COO_NNZ_TH 12       /* COO_NNZ_TH */
#define TILESPMV_PREFETCH_SMEM_TH 4  /* PREFETCH_SMEM_TH: tiles per row-block chunk */

/* Per-tile format tags stored in Tile_matrix.Format (reference src/csr2tile.h:154-319). */
enum {
    TILESPMV_FMT_CSR = 0,
    TILESPMV_FMT_COO = 1,
    TILESPMV_FMT_ELL = 2,
    TILESPMV_FMT_HYB = 3,
    TILESPMV_FMT_DNS = 4,
    TILESPMV_FMT_DNSROW = 5,
    TILESPMV_FMT_DNSCOL = 6
};

/*
 * Tile_matrix — field names, order and types are those of reference src/format.h:3-56
 * (tests and callers read the fields by name).  Semantics after Tile_create are listed in
 * arrays of length tilenum+1 are exclusive prefixes.
 */
typedef struct {
    int tilem;
    int tilen;
    int tilenum;
    MAT_PTR_TYPE *tile_ptr;
    int *tile_columnidx;
    int *tile_nnz;
    char *Format;
    int *blknnz;
    unsigned char *blknnznnz;
    int *dnsrowptr;
    int *dnscolptr;
    char *tilewidth;
    int *csr_offset;
    int *csrptr_offset;
    int *coo_offset;
    int *ell_offset;
    int *hyb_offset;
    int *hyb_coocount;
    int *dns_offset;
    int *dnsrow_offset;
    int *dnscol_offset;
    int *new_coocount;
    MAT_VAL_TYPE *Blockcsr_Val;
    unsigned char *Blockcsr_Ptr;
    unsigned char *csr_compressedIdx;
    int csrsize;
    int csrptrlen;
    MAT_VAL_TYPE *Blockcoo_Val;
    unsigned char *coo_compressed_Idx;
    int coosize;
    MAT_VAL_TYPE *Blockell_Val;
    unsigned char *ell_compressedIdx;
    int ellsize;
    MAT_VAL_TYPE *Blockhyb_Val;
    unsigned char *hybIdx;
    int hybsize;
    int hybellsize;
    int hybcoosize;
    MAT_VAL_TYPE *Blockdense_Val;
    int dnssize;
    MAT_VAL_TYPE *Blockdenserow_Val;
    char *denserowid;
    int dnsrowsize;
    MAT_VAL_TYPE *Blockdensecol_Val;
    char *densecolid;
    int dnscolsize;
    int coototal;
    MAT_VAL_TYPE *deferredcoo_val;
    int *deferredcoo_colidx;
    MAT_PTR_TYPE *deferredcoo_ptr;
} Tile_matrix;

/* ------------------------------------------------------------------------------------------
 * Host preprocessing (kept API).
 * ---------------------------------------------------------------------------------------- */

/* Replaces reference src/csr2tile.h:629-635.  Caller owns the struct, callee mallocs every
 * member array; inputs are borrowed and not modified.  Prints "\n  The number of tile = %i\n"
 * (reference src/csr2tile.h:661).  Output is byte-identical to the reference's. */
void Tile_create(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA,
                 MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA, MAT_VAL_TYPE *csrValA);

/* Same as Tile_create with option bits.  TILESPMV_CREATE_HYB enables the HYB selection rule
 * that is commented out in the shipped reference (src/csr2tile.h:308-316);
 * TILESPMV_CREATE_QUIET suppresses the stdout line. */
#define TILESPMV_CREATE_HYB 1u
#define TILESPMV_CREATE_QUIET 2u
/* TILESPMV_CREATE_CDNA4 (opt-in; default off = the reference's selection, byte-identical Tile_matrix): per-tile format chosen by
 * the bytes THIS engine moves for it (16-value units + entries) instead of the reference's thresholds (dense at 75 % fill, COO up
 * to 12 entries, ELL at row-length variation <= 0.2: src/csr2tile.h:150,159,267-270; COO_NNZ_TH src/common.h:45-47).  The result is
 * still a valid Tile_matrix for tilespmv_cpu and every plan; profiles/r03_selection_cdna4.txt has what it changes. */
#define TILESPMV_CREATE_CDNA4 4u
void Tile_create_ex(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA,
                    const MAT_PTR_TYPE *csrRowPtrA, const int *csrColIdxA,
                    const MAT_VAL_TYPE *csrValA, unsigned flags);

/* Replaces reference src/format.h:58-94: frees the member arrays (all of them, including the
 * four the reference leaks), not the struct. */
void Tile_destroy(Tile_matrix *matrix);

/* Replaces reference src/tilespmv_cpu.h:3-18: builds the row-block schedule
 * (ptroffset1/2[tilenum] caller-allocated; rowblkblock and the three chunk arrays are
 * malloc'd here and owned by the caller), computes y = A*x serially on the host in tile
 * order, compares with y_golden (exact) and prints " Run CPU TileSpMV, errcount = %i\n".
 * Host-side schedule builder + host check only — the GPU path never calls it. */
void tilespmv_cpu(Tile_matrix *matrix, int *ptroffset1, int *ptroffset2, int *rowblkblock,
                  unsigned int **blkcoostylerowidx, int **blkcoostylerowidx_colstart,
                  int **blkcoostylerowidx_colstop, int rowA, int colA, MAT_PTR_TYPE nnzA,
                  MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA, MAT_VAL_TYPE *csrValA,
                  MAT_VAL_TYPE *x, MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden);

/* Replaces reference src/mmio_highlevel.h:593-759.  Returns 0, -1 (open), -2 (banner),
 * -4 (size line).  Entries land in file order; symmetric/hermitian files are mirrored,
 * skew-symmetric are not; pattern -> 1.0; complex keeps the real part. */
int mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr,
                  int **csrColIdx, MAT_VAL_TYPE **csrVal, char *filename);

/* Binary cache of a created Tile_matrix (new: the reference never serialises it; a multi-GB
 * .mtx is otherwise re-parsed and re-tiled on every run).  save: 0 on success, -1 cannot open,
 * -3 short write.  load: callee mallocs every member (free with Tile_destroy); 0 on success,
 * -1 cannot open, -2 not a cache file (or another format version), -3 read error, -5 written by the other value
 * type, -6 corrupt / truncated / stale: the header's counts, the file length, the payload checksum (FNV-1a-64) and the
 * prefix arrays are all checked before a matrix is handed back.  A cache does not record the Tile_create flags: keep
 * caches built with TILESPMV_CREATE_HYB apart from the others (the Format array tells them apart after loading). */
int tilespmv_matrix_save(const Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, const char *path);
int tilespmv_matrix_load(Tile_matrix *matrix, int *rowA, int *colA, MAT_PTR_TYPE *nnzA, const char *path);

/* Parse once (new): binary cache of what mmio_allinone returns.  The reference re-tokenises the text on
 * every run (src/mmio_highlevel.h:648-682; nlpkkt160 is ~4 GB of it).  save: 0, -1 cannot open, -3 short write (the partial
 * file is removed).  load: arrays are malloc'd like mmio_allinone's (caller frees); 0, -1 cannot open, -2 not a CSR cache,
 * -3 read error, -5 other value type, -6 corrupt (length, FNV-1a-64 checksum, row pointer, column range all checked), -7
 * stale: `source_mtx` (may be NULL = do not check) no longer has the size and modification time recorded at save time. */
int tilespmv_csr_save(const char *path, int m, int n, MAT_PTR_TYPE nnz, int isSymmetric,
                      const MAT_PTR_TYPE *csrRowPtr, const int *csrColIdx, const MAT_VAL_TYPE *csrVal,
                      const char *source_mtx);
int tilespmv_csr_load(const char *path, int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric,
                      MAT_PTR_TYPE **csrRowPtr, int **csrColIdx, MAT_VAL_TYPE **csrVal,
                      const char *source_mtx);
/* mmio_allinone with the cache beside it: reads `cache_path` when it is fresh for `filename`, else parses the text and
 * (re)writes the cache.  *from_cache (may be NULL): 1 read from the cache, 0 parsed and saved, -1 parsed, cache not
 * writable.  Return codes of mmio_allinone.  The CLI's `--cache[=prefix]` and bench.py's `--cache DIR` go through this. */
int mmio_allinone_cached(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr,
                         int **csrColIdx, MAT_VAL_TYPE **csrVal, char *filename, const char *cache_path,
                         int *from_cache);
/* Writes a general coordinate Matrix Market file in CSR order (the reference's counterpart: mm_write_mtx_crd,
 * src/mmio.h:605-645), formatted by all host threads; csrVal == NULL writes a pattern file.  0, -1 cannot open, -3 short
 * write. */
int tilespmv_mtx_write(const char *path, int m, int n, MAT_PTR_TYPE nnz, const MAT_PTR_TYPE *csrRowPtr,
                       const int *csrColIdx, const MAT_VAL_TYPE *csrVal);

/* ------------------------------------------------------------------------------------------
 * GPU hot path.
 * ---------------------------------------------------------------------------------------- */

/* Replaces call_tilespmv_cuda, reference src/tilespmv_cuda.h:794-809 (same argument list and
 * meaning; all pointers are HOST pointers; alpha is accepted and ignored like the reference).
 * Uploads the tiled matrix to the current HIP device, runs WARMUP_NUM warm-up and
 * BENCH_REPEAT timed SpMVs (env TILESPMV_WARMUP / TILESPMV_BENCH_REPEAT override 200 / 1000),
 * prints "  CUDA SpMV runtime %4.2f ms, %4.2f GFlops\n\n" (kept verbatim for log parsers,
 * reference :1139) plus one added "  HIP ..." line, appends "file,rowA,colA,nnzA,ms,gflops"
 * to ./results.csv (reference :1142-1147) and copies y back.  Aborts with a message and a
 * non-zero exit status on any HIP error (the reference checks nothing). */
void call_tilespmv_hip(char *filename, Tile_matrix *matrix, int *ptroffset1, int *ptroffset2,
                       int rowblkblock, unsigned int *blkcoostylerowidx,
                       int *blkcoostylerowidx_colstart, int *blkcoostylerowidx_colstop, int rowA,
                       int colA, MAT_PTR_TYPE nnzA, MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA,
                       MAT_VAL_TYPE *csrValA, MAT_VAL_TYPE alpha, MAT_VAL_TYPE *x,
                       MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden);

/* Multi-device form of the same driver (new: the reference is single-GPU, src/main.cu:74; SURVEY.md
 * S8(b) "New").  Same 18 arguments, then the devices to use and what to do with y afterwards.
 * The matrix is cut into `ngpus` nnz-balanced blocks of whole tile-rows (tilespmv_partition_tilerows),
 * one resident plan per device, x replicated; the SpMV needs no exchange.  y_combine_mode:
 *   TILESPMV_Y_SHARDED   every device keeps its own rows (the timed figure of the other modes too);
 *   TILESPMV_Y_ALLGATHER every device receives the other shards' rows by peer copies (xGMI);
 *   TILESPMV_Y_ALLREDUCE ncclAllReduce(sum) of full-length vectors that are zero outside the owner's
 *                        rows (bit-identical to the gather); librccl is dlopen'ed for this mode only.
 * Prints the reference's runtime line for the sharded SpMV plus one "  HIP ..." line that also
 * carries the SpMV+combine time, appends results.csv, returns y on the host (in the combine modes
 * taken from the last device's full-length copy).  A device id may be repeated (several shards on
 * one device) except in all-reduce mode.  Returns 0, or — after a message on stderr and after releasing every device resource it
 * had created — a non-zero status on any error (bad arguments, no such device, HIP / RCCL failure): it never exits the process. */
#define TILESPMV_Y_SHARDED 0
#define TILESPMV_Y_ALLGATHER 1
#define TILESPMV_Y_ALLREDUCE 2
int call_tilespmv_hip_multi(char *filename, Tile_matrix *matrix, int *ptroffset1, int *ptroffset2,
                             int rowblkblock, unsigned int *blkcoostylerowidx,
                             int *blkcoostylerowidx_colstart, int *blkcoostylerowidx_colstop,
                             int rowA, int colA, MAT_PTR_TYPE nnzA, MAT_PTR_TYPE *csrRowPtrA,
                             int *csrColIdxA, MAT_VAL_TYPE *csrValA, MAT_VAL_TYPE alpha,
                             MAT_VAL_TYPE *x, MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden, int ngpus,
                             const int *device_ids, int y_combine_mode);

/*
 * Resident-plan API (new; what call_tilespmv_hip is built from).  A plan owns the device
 * copy of one Tile_matrix (or of one contiguous block of its tile-rows: the multi-GPU shard),
 * re-laid-out for CDNA4.  x / y are DEVICE pointers; stream is a hipStream_t passed as void*.
 */
typedef struct tilespmv_plan tilespmv_plan;

/* How COO tiles (and HYB remainders) are executed — reference has both paths
 * (in-kernel deferred COO: src/tilespmv_cuda.h:462-488; extracted CSR + CSR5: :1011-1029,:1080). */
#define TILESPMV_COO_AUTO 0      /* in-tile with the unit-stream kernel (the fallback is a second launch and never wins there);
                                    by modelled bytes with TILESPMV_KERNEL_DIRECT */
#define TILESPMV_COO_IN_TILE 1   /* COO entries inside the fused launch */
#define TILESPMV_COO_FALLBACK 2  /* very-sparse fallback kernel over the extracted matrix (y += A_coo x), as the reference's CSR5 call */

/* Dense-tile arithmetic. */
#define TILESPMV_DENSE_AUTO 0
#define TILESPMV_DENSE_MFMA 1  /* v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 */
#define TILESPMV_DENSE_VALU 2

/* Fused-kernel generation. */
#define TILESPMV_KERNEL_AUTO 0
#define TILESPMV_KERNEL_DIRECT 1  /* strip-per-16-lanes, one tile at a time (first generation) */
#define TILESPMV_KERNEL_STREAM 2  /* flat, index-addressed unit stream (second generation; default) */

/* Plan options.  VERSIONED: set `size = sizeof(tilespmv_plan_options)` (tilespmv_plan_options_init does, and sets every
 * knob to TILESPMV_KNOB_DEFAULT); a library built against a longer struct gives the fields beyond `size` their defaults,
 * one built against a shorter struct ignores the tail.  opts == NULL = all defaults.
 * The tuning knobs replace what used to travel through the process environment: an unset knob (TILESPMV_KNOB_DEFAULT)
 * takes the value of the environment variable named beside it if that is set — read with getenv at plan creation,
 * never written: the library does not call setenv / unsetenv — and the built-in default otherwise.  Two host threads may
 * create differently tuned plans at the same time (scripts/tsan_host.sh). */
#define TILESPMV_KNOB_DEFAULT (-1)
typedef struct {
    unsigned size;      /* sizeof(tilespmv_plan_options) in the caller's build */
    int coo_mode;       /* TILESPMV_COO_*    (0 = AUTO; env TILESPMV_COO_MODE)   */
    int dense_mode;     /* TILESPMV_DENSE_*  (0 = AUTO; env TILESPMV_DENSE_MODE) */
    int kernel;         /* TILESPMV_KERNEL_* (0 = AUTO; env TILESPMV_KERNEL)     */
    int tilerow_begin;  /* shard: first tile-row (0 for the whole matrix) */
    int tilerow_end;    /* shard: one past the last tile-row (<=0 means tilem) */
    int autotune;       /* != 0 (or env TILESPMV_AUTOTUNE=1): decide the AUTO modes, entry mode, strip size, XCD map and stream cache policy by timing candidates
                           (one candidate plan is resident beside the best one so far: peak device memory = two plans + x + y) */
    /* ---- knobs (TILESPMV_KNOB_DEFAULT = unset) */
    int entry_mode;     /* COO entry lists per 16-lane strip (0), per wavefront (1), per workgroup (2)      TILESPMV_WAVE_COO */
    int entry_ordered;  /* workgroup entry mode: 1 = wavefronts add in turn (bit-reproducible sums), 0 = not  TILESPMV_COO_ORDERED */
    int strip_cost;     /* strip size target in cost units (<= 0: chosen from the shard)                     TILESPMV_STRIP_COST */
    int split_above;    /* tile-rows above this cost are cut into pieces                                     TILESPMV_SPLIT_ABOVE */
    int split_cap;      /* ... and the cap of that threshold in the wavefront / workgroup entry modes        TILESPMV_SPLIT_CAP */
    int xcd_remap;      /* workgroup -> XCD map: 0 round-robin, 2 windows of 8 x xcd_chunk workgroups         TILESPMV_XCD_REMAP */
    int xcd_chunk;      /*                                                                                   TILESPMV_XCD_CHUNK */
    int csr_split;      /* what the device executes for CSR-format tiles: 0 whole tiles in their own pass (first-generation routine, y +=); 1 ELL-style split — the first w
                           entries of every row as w zero-padded 16-value units, the rest as list entries; 2 POOLED units (round 5) — the nonzeros of a tile-row's CSR tiles, COO
                           tiles and HYB remainders pooled in column-major order and cut into units of up to 16 nonzeros inside a 16-column window of x (value + column-offset nibble +
                           row nibble per slot, products scattered into the strip's LDS rows): no padding beyond the last unit of a run of columns, about s_v + 1.3 bytes per nonzero
                           on block-structured / FEM-like matrices; unset: 1 or 2, whichever puts at least 5 % fewer bytes into the streams (2 on shards whose nonzeros sit
                           mostly in ragged CSR tiles, 1 on stencil-like shards whose units share a handful of column patterns); 3 WIDE pooled units — the same pooling with
                           windows of 256 columns (a byte of column offset per slot: s_v + 1.75 bytes per nonzero), taken by the builder where 16-column windows leave >= 4 % of the
                           nonzeros on the entry lists AND a unit's sixteen gathers still touch at most 3.5 lines of x on average (window-shuffled meshes)   TILESPMV_CSR_SPLIT */
    int fix_inline;     /* split tile-rows summed inside the unit kernel (1) or by k_fixup_split (0)         TILESPMV_FIX_INLINE */
    int coo_cost;       /* cost units per COO entry in the strip cutter                                      TILESPMV_COO_COST */
    int coo_heavy_min;  /* entry mode 0: strips with more entries run their list before the unit pipeline    TILESPMV_COO_HEAVY_MIN */
    int coo_piece;      /* entries per piece of a split tile-row                                             TILESPMV_COO_PIECE */
    int strip_even;     /* strips end on multiples of this many units                                        TILESPMV_STRIP_EVEN */
    int wg_strips;      /* workgroup entry mode: 16 (256-thread workgroups) or 32 (512 threads) strips per workgroup  TILESPMV_WG_STRIPS */
    int x_window;       /* stencil-like shards: -1 / unset = brick task order on large 3-D shards, 0 = off, 1 / 2 = brick order wherever grid strides are
                           found.  (Rounds 3-5: 1 also staged the workgroup's x segments in LDS; measured 25 % slower, retired in round 6 together with the
                           slab-paced entry phase and its five `pace*` knobs)                                                 TILESPMV_X_WINDOW */
    int x_stride1;      /* ... tile-rows per grid line (0 / unset: detected from the shard)                  TILESPMV_X_STRIDE1 */
    int x_stride2;      /* ... tile-rows per grid plane (0 / unset: detected; none for 2-D problems)         TILESPMV_X_STRIDE2 */
    int mv_native;      /* tilespmv_plan_spmm: 0 one vector at a time, 1 multi-vector kernel with per-strip entries, 2 multi-vector kernel +
                           entry pass over the merged lists (entry mode 2); unset: by plan and nvec                        TILESPMV_MV_NATIVE */
    int mv_xcd_chunk;   /* XCD window of the multi-vector kernel                                             TILESPMV_MV_XCD_CHUNK */
    int lds_pad;        /* bytes of unused LDS added to every unit-kernel workgroup: fewer resident workgroups per CU  TILESPMV_LDS_PAD */
    int y_store;        /* y stores: 1 streaming (nontemporal), 0 plain; unset: streaming where y is >= 5 % of the launch's bytes  TILESPMV_Y_STORE */
    int desc_dict;      /* unit descriptors: unset = 4 B per unit + a dictionary of column patterns where the shard's units use few distinct
                           patterns AND the 8 bytes per unit are >= 2 % of the streams (stencil-like shards); 1 = wherever the patterns
                           are few; 0 = always the 12-B form.  Pooled plans (csr_split 2): any value but 0 = pattern dictionary where the patterns are few, as ONE
                           4-byte word per unit where window base, pattern id and tile-row fit it (round 6), else as 8-byte pairs; 2 = always pairs;
                           0 = the 20-byte form                                                                                 TILESPMV_DESC_DICT */
    int nt_stream;      /* value / entry-record / dense-tile loads: 1 nontemporal, 0 default cache policy; unset: nontemporal where one SpMV moves more
                           than 400 MB (about 1.6 x the Infinity Cache)                                                             TILESPMV_NT_STREAM */
    int x_panel_kb;     /* column panels (round 4): the merged entry lists of the workgroup entry mode are in column order, so the entries of a column panel (this many
                           KB of x; a power of two) are a run of a list; the plan records where the panels begin.  A panelled launch gives the unit kernel the first
                           x_panel_merge panels and every further run of x_panel_merge panels one more launch that adds its entries (y +=): the kernel boundary is the
                           one cheap chip-wide synchronisation, so all gathers of a pass fall into one slice of x and more of them hit the L2s (scattered matrices
                           with a large x: uniform random 8 M rows 1.00 -> 0.80 ms).  0 = no panels; unset: 2048 on entry-dominated shards whose x is >= 12 MB
                                                                                                                             TILESPMV_X_PANEL_KB */
    int x_panel_merge;  /* ... panels per pass: 0 = whole lists in the unit kernel (unpanelled launch); unset: chosen by timing at plan creation among 0 and the merges
                           that make passes of 4, 8 and 16 MB of x (kept when >= 3 % faster than the unpanelled launch)       TILESPMV_X_PANEL_MERGE */
    int placement_tries; /* where a large plan's blocks land in the card's memory decides between two states 13 % apart on the KKT matrices:
                           at plan creation the plan is timed (5 launches), moved to freshly allocated blocks (allocated BEFORE the old ones are freed) and timed
                           again, up to this many placements (all held until the choice is made); the fastest one seen is kept, and the search ends early once a placement is >= 9 % faster
                           than a slow time two placements agree on, or five placements agree within 1.5 %.  unset: 8 for plans of >= 1 GB, else 1 (= off)
                                                                                                                    TILESPMV_PLACEMENT_TRIES */
    int x_slice_passes; /* column slices pinned to XCDs (round 4): the other use of the recorded panels.  The unit kernel leaves the entry lists alone; per pass one launch of
                           8 x groups workgroups follows in which workgroup b (dispatched to XCD b & 7) takes its group's entries of column slice pass * 8 + (b & 7), so an XCD
                           only ever gathers from its own slice of x, which stays in its L2, and adds the rows it touched to y atomically — the eight partial sums of a row
                           meet in an order that is not fixed: never chosen when entry_ordered = 1.  N > 0 = N passes (8 N slices); 0 = off; unset (and x_panel_merge unset): timed at plan creation
                           beside the panelled forms (1, 2, 4 passes where a slice would be about 1-8 MB), kept when fastest and >= 3 % faster than the plain launch.
                           NOTE: that timed choice can pick this form on large grids where the rule alone would have added in a fixed order — the default plan of a large scattered shard
                           is then NOT bit-reproducible on real-valued data (its facts say so: TILESPMV_INFO_ENTRY_ORDERED = 0, TILESPMV_INFO_X_SLICE_PASSES > 0) and which form it has can differ
                           from run to run and rank to rank; entry_ordered = 1 or deterministic = 1 pins reproducible sums (bench.py reports what that costs per workload)
                                                                                                                    TILESPMV_X_SLICE_PASSES */
    int deterministic;  /* 1: no decision of this plan is taken by a stopwatch and every sum has a plan-fixed order — placement_tries, x_panel_merge, x_slice_passes, pace and autotune
                           that the caller left unset are switched off (as if 1 / 0 / 0 / 0 / 0) and entry_ordered is 1: two plans of the same matrix then have the same layout,
                           the same launch form and give bit-identical y on any data, run after run and rank after rank (the reference's own timing loop never changes the
                           result either, src/tilespmv_cuda.h:1112-1137).  0 / unset: the defaults described above                       TILESPMV_DETERMINISTIC */
    int absorb;         /* round 6: list entries of a COO tile that sit within a few columns of a neighbouring ELL tile (the corner entries of a band / stencil) move into
                           that tile's padding slots — the unit's 16-column window of x is shifted by -3 .. 3 columns — instead of going to the strip's entry list; and a
                           unit whose columns are the previous unit's moved one to the right (consecutive diagonals of a band tile) takes that unit's x one lane up
                           instead of gathering.  unset / 1 = both, where they apply (classic unit plans); 2 = absorbed entries only; 0 = neither                                                    TILESPMV_ABSORB */
    int reserved[1];    /* must be TILESPMV_KNOB_DEFAULT or 0 */
} tilespmv_plan_options;
void tilespmv_plan_options_init(tilespmv_plan_options *opts);
/* "name:offset,..." of every field above as the library was built: lets a binding that mirrors the struct by hand verify
 * its field order (the Python mirror once had four knobs permuted and nothing noticed). */
const char *tilespmv_plan_options_layout(void);

/* Returns 0 on success, non-zero (message on stderr) when no HIP device / extension is
 * usable — there is no CPU fallback behind this entry point.
 * Plan creation allocates, copies and synchronises (it always did); since round 4 it may also TIME a few launches of the finished plan on the default stream with scratch
 * x / y of its own: plans of >= 1 GB try up to `placement_tries` memory placements, shards with column panels recorded choose the panels per pass, opt-in paced plans
 * calibrate their timetable.  Every such choice has a knob that fixes it (placement_tries = 1, x_panel_merge + x_slice_passes, pace_period_us) — a fixed choice times nothing;
 * `deterministic = 1` fixes them all.  Since round 5 the value stream's final layout is written by a kernel (the emitted values are uploaded to a scratch buffer first): peak device
 * memory during creation = the plan + one more copy of its unit values (freed before the call returns); TILESPMV_ENCODE_ON_HOST=1 keeps that pass on the host. */
int tilespmv_plan_create(tilespmv_plan **plan, const Tile_matrix *matrix, int rowA, int colA,
                         MAT_PTR_TYPE nnzA, const tilespmv_plan_options *opts);
void tilespmv_plan_destroy(tilespmv_plan *plan);

/* Preprocessing on the device (new; SURVEY.md S8 f1 "device-side"; replaces the reference's Tile_create + upload, src/csr2tile.h:629-1020 + src/tilespmv_cuda.h:828-1057, end to end).
 *
 * Tile_create_device: the host Tile_create computed by kernels — the CSR arrays go up, every member array of the Tile_matrix comes back, byte for byte what Tile_create /
 * Tile_create_ex builds (flags: TILESPMV_CREATE_QUIET, TILESPMV_CREATE_CDNA4, and since round 6 TILESPMV_CREATE_HYB — width search src/csr2tile.h:279-306, pack :505-548, index bytes
 * :984-1008 as per-tile device code).  Returns 0, -1 no HIP device (there is no CPU fallback behind this entry point: call Tile_create), -2 an offset leaves the int32 range
 * of Tile_matrix, -3 HIP error / out of device memory.
 *
 * tilespmv_plan_create_from_csr: CSR in, resident plan out, nothing but the CSR arrays crossing the bus — the tiled matrix is built on the device and stays there, the stages of the
 * plan builder that touch every nonzero run as kernels calling the same per-tile functions as the host builder, so the plan is the one tilespmv_plan_create makes from
 * Tile_create's output (same streams, same y).  Not every option has a device path: returns -4 (and builds nothing) for the first-generation kernel, the CSR fallback
 * mode and whole CSR tiles (csr_split = 0) — use Tile_create + tilespmv_plan_create for those (HYB tiles, TILESPMV_CREATE_HYB in `create_flags`, are served since round 6).  autotune = 1 is served: every candidate plan is built from the one device-resident tiled
 * matrix (the CSR-fallback candidate, which has no device path, is not among them).  Other return codes as above.
 * Peak device memory during the call: the CSR arrays + the tiled matrix + the sort's key buffers (about 40 bytes per nonzero in fp64) beside the plan. */
int Tile_create_device(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, const MAT_PTR_TYPE *csrRowPtrA,
                       const int *csrColIdxA, const MAT_VAL_TYPE *csrValA, unsigned flags);
int tilespmv_plan_create_from_csr(tilespmv_plan **plan, int rowA, int colA, MAT_PTR_TYPE nnzA, const MAT_PTR_TYPE *csrRowPtrA,
                                  const int *csrColIdxA, const MAT_VAL_TYPE *csrValA, unsigned create_flags,
                                  const tilespmv_plan_options *opts);
/* The same with the CSR arrays already in DEVICE memory (a solver that assembles on the GPU, a framework's CSR tensor): nothing crosses the bus but the few numbers the host decides on.
 * The row pointer starts at 0; the arrays are borrowed for the duration of the call and not modified. */
int tilespmv_plan_create_from_device_csr(tilespmv_plan **plan, int rowA, int colA, MAT_PTR_TYPE nnzA, const MAT_PTR_TYPE *d_csrRowPtrA,
                                         const int *d_csrColIdxA, const MAT_VAL_TYPE *d_csrValA, unsigned create_flags,
                                         const tilespmv_plan_options *opts);

/* Values: x (and the matrix values) must be FINITE.  Zero-padded payload is multiplied by real x entries — the padding
 * slots of ELL / HYB tiles exactly as in the reference (src/tilespmv_cpu.h:173-192 walks all `width` slots), and in
 * this engine also CSR tiles re-expressed as units, dense tiles, and the clamped reads of a partial last column block —
 * so an Inf or NaN in x can reach rows that store no entry in that column (0 * Inf = NaN).  One more carrier, outside the tile: the merged entry
 * lists (wavefront / workgroup entry modes, CSR fallback) pad a 64-record chunk that had to be closed early — its columns span 2^(32 - dest_bits)
 * (2^20 ... 2^23) or more, or, in slab-paced plans, the local part of a list ends — with null records (value 0, destination = the group's first
 * row, column = the chunk's first column): they add 0 * x[that column] to that row (tests/test_gpu_configs.py::test_non_finite_x_reaches_only_what_the_header_says).
 *
 * y[16*tilerow_begin .. 16*tilerow_end) = A_shard * x.  d_x has colA elements, d_y points at
 * element 0 of the FULL-length y (the shard writes only its own rows).  Asynchronous on
 * `stream`, no allocation or synchronisation inside (safe to capture into a hipGraph).  One plan
 * must not execute on two streams at the same time: split tile-rows use per-plan scratch slots
 * and counters.  Returns a hipError_t value (0 = success). */
int tilespmv_plan_spmv(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y,
                       void *stream);

/* `count` back-to-back SpMVs on `stream` (same as calling tilespmv_plan_spmv `count` times; saves the
 * caller's per-call overhead when one SpMV takes tens of microseconds).  Returns a hipError_t value. */
int tilespmv_plan_spmv_n(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y, void *stream, int count);

/* Multi-vector form (SpMM; new): Y[rows][nvec] = A_shard * X[colA][nvec], X and Y
 * row-major (the nvec values of one row are contiguous) and 16-byte aligned, nvec in {1, 2, 4, 8}.
 * The matrix is streamed once for all nvec right-hand sides.  d_Y points at row 0 of the full-length
 * Y.  Unit-stream plans with in-tile COO and split CSR tiles (the defaults) run the native multi-vector
 * kernels; plans built with TILESPMV_COO_FALLBACK, TILESPMV_KERNEL_DIRECT or TILESPMV_CSR_SPLIT=0 go one
 * right-hand side at a time through their own SpMV (column gathered / scattered by two small kernels).
 * Returns hipErrorInvalidValue (1) for other nvec / misaligned pointers. */
#define TILESPMV_MAX_NVEC 8
int tilespmv_plan_spmm(tilespmv_plan *plan, const MAT_VAL_TYPE *d_X, MAT_VAL_TYPE *d_Y, int nvec,
                       void *stream);
double tilespmv_plan_time_spmm(tilespmv_plan *plan, const MAT_VAL_TYPE *d_X, MAT_VAL_TYPE *d_Y,
                               int nvec, void *stream, int warmup, int reps);
/* Plans without a native multi-vector kernel (CSR fallback, first-generation kernel, whole CSR tiles; or mv_native = 0) run one right-hand side at a time
 * on transposed copies of X and Y that live with the plan.  tilespmv_plan_reserve_spmm allocates them for up to `nvec`
 * right-hand sides (hipMalloc: synchronises) so that tilespmv_plan_spmm itself never allocates — call it before capturing
 * tilespmv_plan_spmm into a hipGraph.  Without it the first such tilespmv_plan_spmm call allocates (and would fail under
 * stream capture).  Returns a hipError_t value. */
int tilespmv_plan_reserve_spmm(tilespmv_plan *plan, int nvec);

/* Test / diagnostic entry (new): builds the plan's device layout ON THE HOST ONLY — no HIP call, works without a GPU — with
 * exactly the code tilespmv_plan_create runs, hashes every stream it would upload (FNV-1a-64 over element counts and bytes,
 * in upload order) and checks that the packed entry lists decode back to their entries.  Returns 0 and the digest (and the
 * plan facts, `info` may be NULL), or the error tilespmv_plan_create would return.  Used by the CPU test-suite and by the
 * ThreadSanitizer driver (two threads building differently tuned layouts); it is not part of any compute path. */
int tilespmv_plan_layout_digest(const Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA,
                                const tilespmv_plan_options *opts, unsigned long long *digest,
                                long long *info /* [TILESPMV_INFO_COUNT] or NULL */);

/* Same host-only build, one digest per STAGE of the unit-stream layout builder (hip_plan_stream.hip), so that a test can say which
 * knob is allowed to change which stage (`stage_digests` = TILESPMV_STAGE_COUNT values; all 0 for first-generation plans). */
enum {
    TILESPMV_STAGE_COUNT_ROWS = 0,    /* per tile-row: units / entries / whole tiles / dense tiles, cost */
    TILESPMV_STAGE_CHOOSE = 1,        /* strip size, entry mode, strips per workgroup, ordered adds, brick order */
    TILESPMV_STAGE_CUT = 2,           /* strips and pieces -> task records */
    TILESPMV_STAGE_EMIT = 3,          /* unit descriptors + values, entry triples, whole-tile and dense payload (tile order) */
    TILESPMV_STAGE_ORDER = 4,         /* task order: linear / brick (+ x windows) */
    TILESPMV_STAGE_ENCODE = 5,        /* value groups per task, 12-B descriptors or 4-B words + dictionary */
    TILESPMV_STAGE_ENTRIES = 6,       /* merged, column-ordered, packed entry lists */
    TILESPMV_STAGE_FINISH = 7,        /* byte model and cache-policy flags */
    TILESPMV_STAGE_COUNT = 8
};
int tilespmv_plan_layout_stages(const Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA,
                                const tilespmv_plan_options *opts, unsigned long long *stage_digests,
                                long long *info /* [TILESPMV_INFO_COUNT] or NULL */);

/* Plan facts for reports: index into `out` by TILESPMV_INFO_*. */
enum {
    TILESPMV_INFO_DEVICE_BYTES = 0,   /* bytes of the resident plan */
    TILESPMV_INFO_STREAM_BYTES = 1,   /* bytes one SpMV must read/write at least (plan model) */
    TILESPMV_INFO_NNZ = 2,            /* true nonzeros covered by the shard */
    TILESPMV_INFO_ROWS = 3,
    TILESPMV_INFO_TILES = 4,
    TILESPMV_INFO_COO_MODE = 5,       /* resolved mode */
    TILESPMV_INFO_DENSE_MODE = 6,
    TILESPMV_INFO_KERNEL = 7,
    TILESPMV_INFO_NUM_TASKS = 8,
    TILESPMV_INFO_NUM_SPLIT_ROWS = 9,
    TILESPMV_INFO_FALLBACK_NNZ = 10,  /* nonzeros executed by the CSR fallback kernel */
    TILESPMV_INFO_BUILD_US = 11,      /* host time of the re-layout (plan build), microseconds */
    TILESPMV_INFO_UPLOAD_US = 12,     /* hipMalloc + hipMemcpy time of the plan's streams, microseconds */
    TILESPMV_INFO_ENTRY_MODE = 13,    /* COO entry lists run per strip (0), per wavefront (1), per workgroup (2) */
    TILESPMV_INFO_ENTRY_ORDERED = 14, /* 1: the order of the additions is fixed by the plan (bit-reproducible y: every launch of this plan, and every plan of the same facts, host- or device-built —
                                         column-panel passes and split tile-rows included; tests/test_gpu_parity.py::test_panelled_plans_with_split_rows_sum_in_a_fixed_order, scripts/reproducibility_sweep.py) */
    TILESPMV_INFO_STRIP_COST = 15,    /* strip size target the plan was cut with */
    TILESPMV_INFO_WG_STRIPS = 16,     /* strips per workgroup of the unit kernel (16 or 32) */
    TILESPMV_INFO_LIST_ENTRIES = 17,      /* nonzeros on the strips' entry lists (COO tiles, HYB / CSR-tile remainders) after the ones that fit the padding of a neighbouring ELL unit moved there
                                             (tilespmv_plan_options.absorb); unit-stream plans, 0 otherwise.  (Until round 6: a retired fact that was always 0) */
    TILESPMV_INFO_DERIVED_UNITS = 18,     /* classic unit plans: units that take their x from the previous unit, one lane up, instead of gathering (the consecutive diagonals of a band /
                                             stencil tile: csrc/plan_tile_ops.h; tilespmv_plan_options.absorb = 2 switches them off).  (Until round 6: a retired fact that was always 0) */
    TILESPMV_INFO_BRICK_ORDER = 19,       /* 1: the strips were regrouped into bricks of the grid (stencil-like shard) */
    TILESPMV_INFO_DESC_BYTES = 20,        /* bytes per unit descriptor in HBM: 12, or 4 (column-pattern dictionary); pooled plans 20, or 8 (pattern dictionary); wide pooled plans 28 */
    TILESPMV_INFO_NT_STREAM = 21,         /* 1: the unit kernel reads the value / entry-record streams with nontemporal loads */
    TILESPMV_INFO_RETIRED_22 = 22,        /* always 0 (slab-paced entry phase, retired in round 6) */
    TILESPMV_INFO_RETIRED_23 = 23,        /* always 0 */
    TILESPMV_INFO_PLACEMENT_TRIES = 24,   /* arena placements timed at plan creation (large plans; 0 / 1 = the first one was kept) */
    TILESPMV_INFO_RETIRED_25 = 25,        /* always 0 */
    TILESPMV_INFO_X_PANELS = 26,          /* column panels of the entry lists = launches of the entry part (1 = not panelled) */
    TILESPMV_INFO_X_PANEL_MERGE = 27,     /* recorded panels per pass of the panelled launch (0 = whole lists in the unit kernel) */
    TILESPMV_INFO_SCATTERED_ENTRIES = 28, /* workgroup entry mode: list entries whose column lies more than 2,048 columns outside their group's own rows — the gathers that
                                             no neighbour shares (the chip resolves about 59 G of those per second from a table that misses the L2s: profiles/r04_gather_granule.txt) */
    TILESPMV_INFO_X_SLICE_PASSES = 29,    /* column slices pinned to XCDs: launches of the sliced entry part (0 = not used); 8 x this many slices of x */
    TILESPMV_INFO_CSR_FORM = 30,          /* what CSR-format tiles became: 0 whole tiles (own pass), 1 ELL-style split (units + list entries), 2 pooled units, 3 wide pooled units (256-column windows) */
    TILESPMV_INFO_TIMED_CHOICES_US = 31,  /* microseconds of plan creation spent TIMING candidates (placement retry, column panels / slices); part of build_us; 0 = nothing was timed */
    TILESPMV_INFO_DEVICE_BUILD = 32,      /* 1: built by tilespmv_plan_create_from_csr (Tile_create, COUNT / EMIT / ENCODE on the device) */
    TILESPMV_INFO_TILE_CREATE_US = 33,    /* ... microseconds of its device Tile_create, the upload of the CSR arrays included (0 otherwise) */
    TILESPMV_INFO_COUNT = 34
};
void tilespmv_plan_info(const tilespmv_plan *plan, long long *out /* [TILESPMV_INFO_COUNT] */);
/* Test / audit aid: one (member offset in the plan object, bytes, FNV-1a-64 of the bytes read back from the device) triple per stream of the plan, at most max_streams of them
 * written; returns the number of streams (or -3 on a HIP error).  Two plans of one matrix and one option set — one built from a host Tile_matrix, one by
 * tilespmv_plan_create_from_csr — must agree triple for triple (tests/test_gpu_device_build.py). */
long long tilespmv_plan_stream_digests(const tilespmv_plan *plan, unsigned long long *out /* [3 * max_streams] */, long long max_streams);

/* Times `reps` back-to-back SpMVs on `stream` with hipEvents recorded on that stream (after
 * `warmup` untimed ones) and returns the mean milliseconds per SpMV, or a negative value on
 * error.  Used by bench.py for the live per-launch figure. */
double tilespmv_plan_time(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y,
                          void *stream, int warmup, int reps);

/* The reference's own timing protocol as a C loop (src/tilespmv_cuda.h:1112-1137): wall clock (gettimeofday) around one
 * launch + stream synchronize, averaged over `reps`; milliseconds per SpMV, negative on error.  call_tilespmv_hip prints
 * this number in the reference's "CUDA SpMV runtime" line; bench.py reports it as reference_style_timing. */
double tilespmv_plan_time_reference_style(tilespmv_plan *plan, const MAT_VAL_TYPE *d_x, MAT_VAL_TYPE *d_y,
                                          void *stream, int reps);

/* Multi-GPU helper: nnz-balanced contiguous tile-row partition (new; the reference is
 * single-GPU, src/main.cu:74).  Writes nparts+1 tile-row boundaries. */
void tilespmv_partition_tilerows(const Tile_matrix *matrix, int nparts, int *bounds);

/* Library facts. */
int tilespmv_sizeof_value(void);    /* 8 or 4 */

/* ---- Permuted-numbering plans (new, round 6; no reference counterpart: the reference multiplies in the numbering of its file, src/main.cu:59-110).
 * For callers that STAY in the permuted numbering — a solver: x is permuted once at entry, K products run on plan-ordered vectors, y is un-permuted once at exit
 * (tilespmv_amd/halo.py HaloSpMV(reorder=True) / cg).  Around ONE product the two permutations cost what the better numbering saves (profiles/r05_rcm_probe.txt).
 *   tilespmv_reorder_rcm   reverse Cuthill-McKee on the symmetrised pattern of the leading n x n block (host, deterministic; perm[new] = old)
 *   tilespmv_csr_permute   B = P A P^T; columns >= n (a rank's halo columns) stay; every row of B comes out in ascending column order (the reference's dense-row / dense-col
 *                          tiles assume that: src/csr2tile.h:586,600-605); val / out_val may be NULL
 *   tilespmv_permute_vector  device: scatter = 0: out[i] = in[perm[i]] (into plan order), scatter = 1: out[perm[i]] = in[i] (back); asynchronous on `stream`
 * The plan of B is created like any other (Tile_create + tilespmv_plan_create, or tilespmv_plan_create_from_csr). */
int tilespmv_reorder_rcm(int n, const MAT_PTR_TYPE *csrRowPtr, const int *csrColIdx, int *perm /* [n] */);
int tilespmv_csr_permute(int n, const MAT_PTR_TYPE *csrRowPtr, const int *csrColIdx, const MAT_VAL_TYPE *csrVal, const int *perm,
                         MAT_PTR_TYPE *outRowPtr /* [n + 1] */, int *outColIdx, MAT_VAL_TYPE *outVal);
long long tilespmv_csr_bandwidth(int n, const MAT_PTR_TYPE *csrRowPtr, const int *csrColIdx);   /* max |i - j| over the leading n x n block */
int tilespmv_permute_vector(const MAT_VAL_TYPE *d_in, MAT_VAL_TYPE *d_out, const int *d_perm, long long n, int scatter, void *stream);   /* hipError_t value */
const char *tilespmv_version(void);
int tilespmv_device_count(void);    /* 0 when no HIP device is visible */

#ifdef __cplusplus
}
#endif
#endif /* TILESPMV_H_ */
