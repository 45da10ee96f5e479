// host_matrix_io.cpp — binary cache of a created Tile_matrix (new; the reference never serialises
// its format, src/format.h:3-56, so large inputs are re-parsed and re-tiled on every run: SURVEY §8 f2).
// Layout: 8-byte magic (format version 002), value size, rowA, colA, nnzA, the 14 scalar fields, the two id-array
// lengths, the payload byte count and an FNV-1a-64 of the payload, then every member array in declaration order with
// the element counts of SURVEY.md Appendix A (derived from the scalars).  A load checks all of it before returning a
// matrix: counts, file length, checksum, and the prefix arrays the plan builder and tilespmv_cpu index with.
#include <unistd.h>

#include <string>

#include "host_util.h"

namespace {

const char MAGIC[8] = {'T', 'S', 'P', 'M', 'V', '0', '0', '2'};

unsigned long long fnv1a(const void *p, size_t n, unsigned long long h)
{
    const unsigned char *b = (const unsigned char *)p;
    // 8 bytes per step keeps a multi-GB cache cheap to verify; the tail goes byte by byte
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { unsigned long long w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001B3ull; }
    for (; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

bool monotone(const int *a, long long n, long long last)   // exclusive prefix: starts at 0, never decreases, ends at `last`
{
    if (n <= 0) return false;
    if (a[0] != 0 || a[n - 1] != last) return false;
    for (long long i = 1; i < n; i++) if (a[i] < a[i - 1]) return false;
    return true;
}

struct Field { void **ptr; size_t elem; long long count; };

// Element counts of every member array after Tile_create.
std::vector<Field> fields_of(Tile_matrix *T, int rowA, long long n_dnsrow, long long n_dnscol)
{
    const long long n = T->tilenum, n1 = n + 1, sv = sizeof(tilespmv::val_t);
    return {
        {(void **)&T->tile_ptr, 4, (long long)T->tilem + 1}, {(void **)&T->tile_columnidx, 4, n}, {(void **)&T->tile_nnz, 4, n1},
        {(void **)&T->Format, 1, n}, {(void **)&T->blknnz, 4, n1}, {(void **)&T->blknnznnz, 1, n1},
        {(void **)&T->dnsrowptr, 4, n1}, {(void **)&T->dnscolptr, 4, n1}, {(void **)&T->tilewidth, 1, n},
        {(void **)&T->csr_offset, 4, n1}, {(void **)&T->csrptr_offset, 4, n1}, {(void **)&T->coo_offset, 4, n1},
        {(void **)&T->ell_offset, 4, n1}, {(void **)&T->hyb_offset, 4, n1}, {(void **)&T->hyb_coocount, 4, n1},
        {(void **)&T->dns_offset, 4, n1}, {(void **)&T->dnsrow_offset, 4, n1}, {(void **)&T->dnscol_offset, 4, n1},
        {(void **)&T->new_coocount, 4, n1},
        {(void **)&T->Blockcsr_Val, (size_t)sv, T->csrsize}, {(void **)&T->Blockcsr_Ptr, 1, T->csrptrlen},
        {(void **)&T->csr_compressedIdx, 1, ((long long)T->csrsize + 1) / 2},
        {(void **)&T->Blockcoo_Val, (size_t)sv, T->coosize}, {(void **)&T->coo_compressed_Idx, 1, T->coosize},
        {(void **)&T->Blockell_Val, (size_t)sv, T->ellsize}, {(void **)&T->ell_compressedIdx, 1, ((long long)T->ellsize + 1) / 2},
        {(void **)&T->Blockhyb_Val, (size_t)sv, (long long)T->hybellsize + T->hybcoosize},
        {(void **)&T->hybIdx, 1, ((long long)T->hybellsize + 1) / 2 + T->hybcoosize},
        {(void **)&T->Blockdense_Val, (size_t)sv, T->dnssize},
        {(void **)&T->Blockdenserow_Val, (size_t)sv, T->dnsrowsize}, {(void **)&T->denserowid, 1, n_dnsrow},
        {(void **)&T->Blockdensecol_Val, (size_t)sv, T->dnscolsize}, {(void **)&T->densecolid, 1, n_dnscol},
        {(void **)&T->deferredcoo_val, (size_t)sv, T->coototal}, {(void **)&T->deferredcoo_colidx, 4, T->coototal},
        {(void **)&T->deferredcoo_ptr, 4, (long long)rowA + 1},
    };
}

int *scalars_of(Tile_matrix *T, int i)
{
    int *s[] = {&T->tilem, &T->tilen, &T->tilenum, &T->csrsize, &T->csrptrlen, &T->coosize, &T->ellsize, &T->hybsize,
                &T->hybellsize, &T->hybcoosize, &T->dnssize, &T->dnsrowsize, &T->dnscolsize, &T->coototal};
    return s[i];
}

}  // namespace

extern "C" int tilespmv_matrix_save(const Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA, const char *path)
{
    Tile_matrix *T = const_cast<Tile_matrix *>(matrix);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int head[4] = {(int)sizeof(tilespmv::val_t), rowA, colA, nnzA};
    bool ok = fwrite(MAGIC, 1, 8, f) == 8 && fwrite(head, sizeof(int), 4, f) == 4;
    for (int i = 0; i < 14 && ok; i++) ok = fwrite(scalars_of(T, i), sizeof(int), 1, f) == 1;
    long long extra[2] = {T->tilenum >= 0 ? T->dnsrowptr[T->tilenum] : 0, T->tilenum >= 0 ? T->dnscolptr[T->tilenum] : 0};
    ok = ok && fwrite(extra, sizeof(long long), 2, f) == 2;
    unsigned long long sum[2] = {0, 0xCBF29CE484222325ull};   // payload bytes, FNV-1a-64 of header fields + payload
    sum[1] = fnv1a(head, sizeof(head), sum[1]);
    for (int i = 0; i < 14; i++) sum[1] = fnv1a(scalars_of(T, i), sizeof(int), sum[1]);
    sum[1] = fnv1a(extra, sizeof(extra), sum[1]);
    for (auto &fd : fields_of(T, rowA, extra[0], extra[1]))
        if (fd.count > 0) { sum[0] += (unsigned long long)fd.count * fd.elem; sum[1] = fnv1a(*fd.ptr, (size_t)fd.count * fd.elem, sum[1]); }
    ok = ok && fwrite(sum, sizeof(unsigned long long), 2, f) == 2;
    for (auto &fd : fields_of(T, rowA, extra[0], extra[1])) {
        if (!ok) break;
        if (fd.count > 0) ok = fwrite(*fd.ptr, fd.elem, (size_t)fd.count, f) == (size_t)fd.count;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -3;
}

extern "C" int tilespmv_matrix_load(Tile_matrix *matrix, int *rowA, int *colA, MAT_PTR_TYPE *nnzA, const char *path)
{
    memset(matrix, 0, sizeof(*matrix));
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[8]; int head[4];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, MAGIC, 8) != 0 || fread(head, sizeof(int), 4, f) != 4) { fclose(f); return -2; }
    if (head[0] != (int)sizeof(tilespmv::val_t)) { fclose(f); return -5; }  // written by the library of the other value type
    bool ok = true;
    for (int i = 0; i < 14 && ok; i++) ok = fread(scalars_of(matrix, i), sizeof(int), 1, f) == 1;
    long long extra[2] = {0, 0};
    ok = ok && fread(extra, sizeof(long long), 2, f) == 2;
    unsigned long long sum[2] = {0, 0};
    ok = ok && fread(sum, sizeof(unsigned long long), 2, f) == 2;
    if (!ok) { fclose(f); memset(matrix, 0, sizeof(*matrix)); return -3; }
    // ---- the header must describe a possible matrix before anything is allocated from it
    Tile_matrix *T = matrix;
    bool sane = head[1] >= 0 && head[2] >= 0 && head[3] >= 0 && extra[0] >= 0 && extra[1] >= 0;
    for (int i = 0; i < 14 && sane; i++) sane = *scalars_of(T, i) >= 0;
    sane = sane && T->tilem == (head[1] + 15) / 16 && T->tilen == (head[2] + 15) / 16 && T->hybsize == T->hybellsize + T->hybcoosize;
    unsigned long long bytes = 0;
    if (sane) for (auto &fd : fields_of(T, head[1], extra[0], extra[1])) bytes += (unsigned long long)std::max<long long>(fd.count, 0) * fd.elem;
    const long here = ftell(f);
    sane = sane && bytes == sum[0] && fseek(f, 0, SEEK_END) == 0 && (unsigned long long)(ftell(f) - here) == bytes && fseek(f, here, SEEK_SET) == 0;
    if (!sane) { fclose(f); memset(matrix, 0, sizeof(*matrix)); return -6; }   // corrupt, truncated or stale cache
    unsigned long long h = 0xCBF29CE484222325ull;
    h = fnv1a(head, sizeof(head), h);
    for (int i = 0; i < 14; i++) h = fnv1a(scalars_of(matrix, i), sizeof(int), h);
    h = fnv1a(extra, sizeof(extra), h);
    for (auto &fd : fields_of(matrix, head[1], extra[0], extra[1])) {
        *fd.ptr = calloc((size_t)std::max<long long>(fd.count, 1), fd.elem);
        if (!*fd.ptr) { ok = false; break; }
        if (fd.count > 0 && fread(*fd.ptr, fd.elem, (size_t)fd.count, f) != (size_t)fd.count) { ok = false; break; }
        if (fd.count > 0) h = fnv1a(*fd.ptr, (size_t)fd.count * fd.elem, h);
    }
    fclose(f);
    if (!ok) { Tile_destroy(matrix); return -3; }
    // ---- payload intact, and the prefix arrays that everything else indexes with are consistent
    const long long n = T->tilenum;
    sane = h == sum[1] && monotone(T->tile_ptr, (long long)T->tilem + 1, n) && monotone(T->tile_nnz, n + 1, T->tile_nnz[n]) &&
           monotone(T->blknnz, n + 1, T->blknnz[n]) && monotone(T->csr_offset, n + 1, T->csrsize) && monotone(T->csrptr_offset, n + 1, T->csrptrlen) &&
           monotone(T->coo_offset, n + 1, T->coosize) && monotone(T->ell_offset, n + 1, T->ellsize) && monotone(T->hyb_offset, n + 1, T->hybsize) &&
           monotone(T->dns_offset, n + 1, T->dnssize) && monotone(T->dnsrow_offset, n + 1, T->dnsrowsize) && monotone(T->dnscol_offset, n + 1, T->dnscolsize) &&
           monotone(T->dnsrowptr, n + 1, extra[0]) && monotone(T->dnscolptr, n + 1, extra[1]) && monotone(T->new_coocount, n + 1, T->coototal) &&
           monotone(T->deferredcoo_ptr, (long long)head[1] + 1, T->coototal) && T->tile_nnz[n] <= head[3];
    for (long long t = 0; t < n && sane; t++) sane = T->tile_columnidx[t] >= 0 && T->tile_columnidx[t] < T->tilen && T->Format[t] >= 0 && T->Format[t] <= 6;
    for (long long i = 0; i < T->coototal && sane; i++) sane = T->deferredcoo_colidx[i] >= 0 && T->deferredcoo_colidx[i] < head[2];
    if (!sane) { Tile_destroy(matrix); return -6; }
    *rowA = head[1]; *colA = head[2]; *nnzA = head[3];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// CSR cache of a parsed .mtx (new): the reference re-parses the text on every run (src/mmio_highlevel.h:648-682); here the
// result of mmio_allinone is written once as a flat binary — magic, value size, m, n, nnz, symmetry flag, size and mtime of
// the source file, FNV-1a-64 of the three arrays — and later runs read three arrays instead of tokenising gigabytes.
// ------------------------------------------------------------------------------------------------
#include <sys/stat.h>

namespace {

const char CSR_MAGIC[8] = {'T', 'S', 'C', 'S', 'R', '0', '0', '2'};

struct SourceId { long long size, mtime_s, mtime_ns; };

bool source_id(const char *path, SourceId *id)
{
    struct stat sb;
    if (!path || stat(path, &sb) != 0) return false;
    id->size = (long long)sb.st_size; id->mtime_s = (long long)sb.st_mtim.tv_sec; id->mtime_ns = (long long)sb.st_mtim.tv_nsec;
    return true;
}

}  // namespace

extern "C" int tilespmv_csr_save(const char *path, int m, int n, MAT_PTR_TYPE nnz, int isSymmetric, const MAT_PTR_TYPE *rowptr,
                                 const int *colidx, const MAT_VAL_TYPE *val, const char *source_mtx)
{
    // written under a name of its own and renamed into place: N ranks that all miss the cache write the same file at the same time (bench.py --cache),
    // and a truncating fopen / a remove() on `path` itself could destroy another rank's finished cache
    const std::string tmp = std::string(path) + "." + std::to_string((long long)getpid()) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return -1;
    int head[5] = {(int)sizeof(tilespmv::val_t), m, n, nnz, isSymmetric};
    SourceId id{0, 0, 0};
    (void)source_id(source_mtx, &id);
    unsigned long long h = 0xCBF29CE484222325ull;
    h = fnv1a(head, sizeof(head), h);
    h = fnv1a(&id, sizeof(id), h);
    h = fnv1a(rowptr, sizeof(MAT_PTR_TYPE) * ((size_t)m + 1), h);
    h = fnv1a(colidx, sizeof(int) * (size_t)nnz, h);
    h = fnv1a(val, sizeof(tilespmv::val_t) * (size_t)nnz, h);
    bool ok = fwrite(CSR_MAGIC, 1, 8, f) == 8 && fwrite(head, sizeof(int), 5, f) == 5 && fwrite(&id, sizeof(id), 1, f) == 1 &&
              fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fwrite(rowptr, sizeof(MAT_PTR_TYPE), (size_t)m + 1, f) == (size_t)m + 1;
    ok = ok && (nnz == 0 || fwrite(colidx, sizeof(int), (size_t)nnz, f) == (size_t)nnz);
    ok = ok && (nnz == 0 || fwrite(val, sizeof(tilespmv::val_t), (size_t)nnz, f) == (size_t)nnz);
    ok = (fclose(f) == 0) && ok;
    ok = ok && rename(tmp.c_str(), path) == 0;
    if (!ok) remove(tmp.c_str());   // never leave half a cache behind — and never touch a finished one
    return ok ? 0 : -3;
}

extern "C" int tilespmv_csr_load(const char *path, int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **rowptr,
                                 int **colidx, MAT_VAL_TYPE **val, const char *source_mtx)
{
    *rowptr = nullptr; *colidx = nullptr; *val = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[8]; int head[5]; SourceId id; unsigned long long want = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, CSR_MAGIC, 8) != 0) { fclose(f); return -2; }
    if (fread(head, sizeof(int), 5, f) != 5 || fread(&id, sizeof(id), 1, f) != 1 || fread(&want, sizeof(want), 1, f) != 1) { fclose(f); return -3; }
    if (head[0] != (int)sizeof(tilespmv::val_t)) { fclose(f); return -5; }
    if (source_mtx) {   // stale: the text file changed (or vanished) since the cache was written
        SourceId now;
        if (!source_id(source_mtx, &now) || now.size != id.size || now.mtime_s != id.mtime_s || now.mtime_ns != id.mtime_ns) { fclose(f); return -7; }
    }
    const long long M = head[1], N = head[2], NZ = head[3];
    const long here = ftell(f);
    const unsigned long long bytes = (unsigned long long)(M + 1) * sizeof(MAT_PTR_TYPE) + (unsigned long long)NZ * (sizeof(int) + sizeof(tilespmv::val_t));
    bool sane = M >= 0 && N >= 0 && NZ >= 0 && fseek(f, 0, SEEK_END) == 0 && (unsigned long long)(ftell(f) - here) == bytes && fseek(f, here, SEEK_SET) == 0;
    if (!sane) { fclose(f); return -6; }
    MAT_PTR_TYPE *rp = (MAT_PTR_TYPE *)malloc(sizeof(MAT_PTR_TYPE) * ((size_t)M + 1));
    int *ci = (int *)malloc(sizeof(int) * (size_t)std::max<long long>(NZ, 1));
    tilespmv::val_t *v = (tilespmv::val_t *)malloc(sizeof(tilespmv::val_t) * (size_t)std::max<long long>(NZ, 1));
    bool ok = rp && ci && v && fread(rp, sizeof(MAT_PTR_TYPE), (size_t)M + 1, f) == (size_t)M + 1 &&
              (NZ == 0 || (fread(ci, sizeof(int), (size_t)NZ, f) == (size_t)NZ && fread(v, sizeof(tilespmv::val_t), (size_t)NZ, f) == (size_t)NZ));
    fclose(f);
    if (ok) {
        unsigned long long h = 0xCBF29CE484222325ull;
        h = fnv1a(head, sizeof(head), h);
        h = fnv1a(&id, sizeof(id), h);
        h = fnv1a(rp, sizeof(MAT_PTR_TYPE) * ((size_t)M + 1), h);
        h = fnv1a(ci, sizeof(int) * (size_t)NZ, h);
        h = fnv1a(v, sizeof(tilespmv::val_t) * (size_t)NZ, h);
        sane = h == want && monotone(rp, M + 1, NZ);
        for (long long i = 0; i < NZ && sane; i++) sane = ci[i] >= 0 && ci[i] < N;
    }
    if (!ok || !sane) { free(rp); free(ci); free(v); return ok ? -6 : -3; }
    *m = (int)M; *n = (int)N; *nnz = (MAT_PTR_TYPE)NZ; *isSymmetric = head[4];
    *rowptr = rp; *colidx = ci; *val = v;
    return 0;
}

// mmio_allinone with a cache beside it: `cache_path` fresh (written from this very file: size and mtime match) -> read it;
// otherwise parse the text and (re)write the cache, best effort.  *from_cache: 1 read from the cache, 0 parsed (and saved),
// -1 parsed but the cache could not be written.  Return codes of mmio_allinone.
extern "C" int mmio_allinone_cached(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr, int **csrColIdx,
                                    MAT_VAL_TYPE **csrVal, char *filename, const char *cache_path, int *from_cache)
{
    if (from_cache) *from_cache = 0;
    if (cache_path && tilespmv_csr_load(cache_path, m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal, filename) == 0) {
        if (from_cache) *from_cache = 1;
        return 0;
    }
    const int rc = mmio_allinone(m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal, filename);
    if (rc != 0 || !cache_path) return rc;
    if (tilespmv_csr_save(cache_path, *m, *n, *nnz, *isSymmetric, *csrRowPtr, *csrColIdx, *csrVal, filename) != 0 && from_cache) *from_cache = -1;
    return 0;
}
