// host_mmio.cpp — Matrix Market coordinate file -> CSR, `mmio_allinone` of the reference
// (src/mmio_highlevel.h:593-759; banner/size parsing src/mmio.h:398-508, :568-603).
//
// Same observable result as the reference (return codes; nnz after mirroring; entries in FILE
// order, each off-diagonal of a symmetric/hermitian file appended to row i and then to row j;
// columns never sorted) but the file is read once into memory and tokenised by hand instead
// of one fscanf per entry, so multi-GB inputs (nlpkkt160) load in seconds, not minutes.
#include <cctype>
#include <string>

#include "host_util.h"

namespace {

struct Cursor {
    const char *p, *end;
    void skip_ws() { while (p < end && isspace((unsigned char)*p)) p++; }
    bool eof() { skip_ws(); return p >= end; }
    bool next_int(long *out)
    {
        skip_ws();
        if (p >= end) return false;
        char *q; long v = strtol(p, &q, 10);
        if (q == p) return false;
        p = q; *out = v; return true;
    }
    bool next_real(double *out)
    {
        skip_ws();
        if (p >= end) return false;
        char *q; double v = strtod(p, &q);
        if (q == p) return false;
        p = q; *out = v; return true;
    }
    std::string line()
    {
        const char *s = p;
        while (p < end && *p != '\n') p++;
        std::string l(s, p);
        if (p < end) p++;
        return l;
    }
};

std::string lower(std::string s) { for (auto &c : s) c = (char)tolower((unsigned char)c); return s; }

}  // namespace

extern "C" int mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr,
                             int **csrColIdx, MAT_VAL_TYPE **csrVal, char *filename)
{
    using namespace tilespmv;
    FILE *f = fopen(filename, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)fsize + 1);
    size_t got = fread(buf.data(), 1, (size_t)fsize, f);
    fclose(f);
    buf[got] = '\0';
    Cursor c{buf.data(), buf.data() + got};

    // banner: "%%MatrixMarket matrix coordinate <field> <symmetry>", fields 2..5 case-insensitive
    char t0[65] = "", t1[65] = "", t2[65] = "", t3[65] = "", t4[65] = "";
    std::string first = c.line();
    if (sscanf(first.c_str(), "%64s %64s %64s %64s %64s", t0, t1, t2, t3, t4) != 5) {
        printf("Could not process Matrix Market banner.\n");
        return -2;
    }
    const std::string object = lower(t1), layout = lower(t2), field = lower(t3), symmetry = lower(t4);
    const bool known_field = field == "real" || field == "complex" || field == "pattern" || field == "integer";
    const bool known_sym = symmetry == "general" || symmetry == "symmetric" || symmetry == "hermitian" || symmetry == "skew-symmetric";
    if (strncmp(t0, "%%MatrixMarket", 14) != 0 || object != "matrix" || (layout != "coordinate" && layout != "array") ||
        !known_field || !known_sym) {
        printf("Could not process Matrix Market banner.\n");
        return -2;
    }
    const bool mirror = symmetry == "symmetric" || symmetry == "hermitian";  // skew-symmetric is not mirrored

    // size line: first non-comment line; if it does not hold three ints, keep scanning tokens
    long M = 0, N = 0, NZ = 0;
    std::string l;
    do {
        if (c.p >= c.end) return -4;
        l = c.line();
    } while (!l.empty() && l[0] == '%');
    if (sscanf(l.c_str(), "%ld %ld %ld", &M, &N, &NZ) != 3) {
        if (!c.next_int(&M) || !c.next_int(&N) || !c.next_int(&NZ)) return -4;
    }

    std::vector<int> ri((size_t)NZ), ci((size_t)NZ);
    std::vector<val_t> vv((size_t)NZ);
    int *count = zalloc<int>((size_t)M + 1);
    const int kind = field == "real" ? 0 : field == "complex" ? 1 : field == "integer" ? 2 : 3;
    for (long i = 0; i < NZ; i++) {
        long a = 0, b = 0, iv = 0; double v = 1.0, im = 0.0;
        if (!c.next_int(&a) || !c.next_int(&b)) { a = b = 0; }
        if (kind == 0) c.next_real(&v);
        else if (kind == 1) { c.next_real(&v); c.next_real(&im); }
        else if (kind == 2) { c.next_int(&iv); v = (double)iv; }
        a--; b--;
        if (a < 0 || a >= M || b < 0 || b >= N) {
            fprintf(stderr, "mmio_allinone: entry %ld (%ld,%ld) outside %ld x %ld in %s\n", i, a + 1, b + 1, M, N, filename);
            free(count);
            return -4;
        }
        count[a]++;
        ri[(size_t)i] = (int)a; ci[(size_t)i] = (int)b; vv[(size_t)i] = (val_t)v;
    }
    if (mirror)
        for (long i = 0; i < NZ; i++)
            if (ri[(size_t)i] != ci[(size_t)i]) {
                if (ci[(size_t)i] >= M) { fprintf(stderr, "mmio_allinone: symmetric file is not square\n"); free(count); return -4; }
                count[ci[(size_t)i]]++;
            }
    exclusive_scan_checked(count, M + 1, "nnz");
    const int total = count[M];
    int *rowptr = zalloc<int>((size_t)M + 1);
    memcpy(rowptr, count, sizeof(int) * ((size_t)M + 1));
    int *cols = zalloc<int>((size_t)total);
    val_t *vals = zalloc<val_t>((size_t)total);
    int *fill = count;  // reuse as write cursors
    for (long i = 0; i < NZ; i++) {
        const int a = ri[(size_t)i], b = ci[(size_t)i];
        int p = fill[a]++;
        cols[p] = b; vals[p] = vv[(size_t)i];
        if (mirror && a != b) { p = fill[b]++; cols[p] = a; vals[p] = vv[(size_t)i]; }
    }
    free(count);
    *m = (int)M; *n = (int)N; *nnz = total; *isSymmetric = mirror ? 1 : 0;
    *csrRowPtr = rowptr; *csrColIdx = cols; *csrVal = vals;
    return 0;
}
