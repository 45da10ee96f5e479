// host_matrix_io.cpp — binary cache of a created Tile_matrix (new; the reference never serialises
// its format, src/format.h:3-56, so large inputs are re-parsed and re-tiled on every run: SURVEY §8 f2).
// Layout: 8-byte magic, value size, rowA, colA, nnzA, the 14 scalar fields, then every member array in
// declaration order with the element counts of SURVEY.md Appendix A (derived from the scalars).
#include "host_util.h"

namespace {

const char MAGIC[8] = {'T', 'S', 'P', 'M', 'V', '0', '0', '1'};

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
    if (ok)
        for (auto &fd : fields_of(matrix, head[1], extra[0], extra[1])) {
            *fd.ptr = calloc((size_t)std::max<long long>(fd.count, 1), fd.elem);
            if (!*fd.ptr) { ok = false; break; }
            if (fd.count > 0 && fread(*fd.ptr, fd.elem, (size_t)fd.count, f) != (size_t)fd.count) { ok = false; break; }
        }
    fclose(f);
    if (!ok) { Tile_destroy(matrix); return -3; }
    *rowA = head[1]; *colA = head[2]; *nnzA = head[3];
    return 0;
}
