// host_mmio.cpp — Matrix Market coordinate file -> CSR, `mmio_allinone` of the reference
// (src/mmio_highlevel.h:593-759; banner/size parsing src/mmio.h:398-508, :568-603).
//
// Same observable result as the reference (return codes; nnz after mirroring; entries in FILE
// order, each off-diagonal of a symmetric/hermitian file appended to row i and then to row j;
// columns never sorted) but the file is mmap'ed (no second copy of a multi-GB text) and the entry
// lines are parsed by all host threads in newline-aligned chunks with a hand tokenizer (exact
// decimal fast path, strtod for everything else), instead of one fscanf per entry.  A file whose
// entries do not sit one per line (legal for fscanf) takes the sequential token-stream path.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <string>

#include <cmath>

#include "host_util.h"

namespace {

struct Cursor {
    const char *p, *end;
    void skip_ws() { while (p < end && isspace((unsigned char)*p)) p++; }
    bool eof() { skip_ws(); return p >= end; }
    // bounded (the mapping is not NUL-terminated): strtol / strtod never see the file directly
    bool next_int(long *out)
    {
        skip_ws();
        const char *q = p;
        bool neg = false;
        if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
        if (q >= end || *q < '0' || *q > '9') return false;
        long v = 0;
        while (q < end && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); q++; }
        p = q; *out = neg ? -v : v; return true;
    }
    bool next_real(double *out)
    {
        skip_ws();
        char tmp[128];
        size_t len = 0;
        while (p + len < end && !isspace((unsigned char)p[len]) && len + 1 < sizeof(tmp)) { tmp[len] = p[len]; len++; }
        tmp[len] = 0;
        char *q; double v = strtod(tmp, &q);
        if (q == tmp) return false;
        p += q - tmp; *out = v; return true;
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

// ---- per-line fast parser -------------------------------------------------------------------
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

inline bool fast_int(const char *&p, const char *end, long *out)
{
    while (p < end && is_blank(*p)) p++;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    if (p >= end || *p < '0' || *p > '9') return false;
    long v = 0;
    int nd = 0;
    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; if (++nd > 18) return false; }
    if (p < end && !is_blank(*p) && *p != '\n') return false;  // "12.5" or "1e3" where an integer is expected
    *out = neg ? -v : v;
    return true;
}

// Correctly rounded for <= 15 significant digits and |exponent| <= 22 (mantissa and power of ten are
// both exact doubles, one rounding in the multiply / divide); anything else goes through strtod.
inline bool fast_real(const char *&p, const char *end, double *out)
{
    static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    while (p < end && is_blank(*p)) p++;
    const char *start = p;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    unsigned long long mant = 0;
    int nd = 0, frac = 0, any = 0;
    while (p < end && *p >= '0' && *p <= '9') { if (mant || *p != '0') { mant = mant * 10 + (unsigned)(*p - '0'); nd++; } any = 1; p++; if (nd > 15) break; }
    if (nd <= 15 && p < end && *p == '.') {
        p++;
        while (p < end && *p >= '0' && *p <= '9') { if (mant || *p != '0') { mant = mant * 10 + (unsigned)(*p - '0'); nd++; } frac++; any = 1; p++; if (nd > 15) break; }
    }
    int e10 = 0;
    bool ok = any && nd <= 15;
    if (ok && p < end && (*p == 'e' || *p == 'E' || *p == 'd' || *p == 'D')) {
        if (*p == 'd' || *p == 'D') ok = false;  // Fortran exponents: let strtod decide exactly like the reference's scanf
        else {
            p++;
            bool eneg = false;
            if (p < end && (*p == '-' || *p == '+')) { eneg = *p == '-'; p++; }
            int ed = 0;
            if (p >= end || *p < '0' || *p > '9') ok = false;
            while (ok && p < end && *p >= '0' && *p <= '9') { e10 = e10 * 10 + (*p - '0'); p++; if (++ed > 4) ok = false; }
            if (eneg) e10 = -e10;
        }
    }
    if (ok && p < end && !is_blank(*p) && *p != '\n') ok = false;
    e10 -= frac;
    if (ok && e10 >= -22 && e10 <= 22) {
        double v = (double)mant;
        v = e10 < 0 ? v / p10[-e10] : v * p10[e10];
        *out = neg ? -v : v;
        return true;
    }
    // slow path: the token as strtod sees it (the buffer ends with a newline or NUL, so it cannot run away)
    char tmp[128];
    size_t len = 0;
    const char *q = start;
    while (q < end && !is_blank(*q) && *q != '\n' && len + 1 < sizeof(tmp)) tmp[len++] = *q++;
    tmp[len] = 0;
    char *stop;
    const double v = strtod(tmp, &stop);
    if (stop == tmp) return false;
    p = start + (stop - tmp);
    if (p < end && !is_blank(*p) && *p != '\n') return false;
    *out = v;
    return true;
}

struct ChunkOut { std::vector<int> r, c; std::vector<tilespmv::val_t> v; bool ok = true; long bad_a = 0, bad_b = 0; bool range_error = false; };

// Parses whole lines of [p, end): one entry per non-blank, non-comment line.
void parse_chunk(const char *p, const char *end, int kind, long M, long N, ChunkOut &o)
{
    while (p < end) {
        while (p < end && is_blank(*p)) p++;
        if (p >= end) break;
        if (*p == '\n') { p++; continue; }
        if (*p == '%') { while (p < end && *p != '\n') p++; continue; }
        long a, b, iv = 0;
        double v = 1.0, im = 0.0;
        bool ok = fast_int(p, end, &a) && fast_int(p, end, &b);
        if (ok && kind == 0) ok = fast_real(p, end, &v);
        else if (ok && kind == 1) ok = fast_real(p, end, &v) && fast_real(p, end, &im);
        else if (ok && kind == 2) { ok = fast_int(p, end, &iv); v = (double)iv; }
        while (ok && p < end && is_blank(*p)) p++;
        if (!ok || (p < end && *p != '\n')) { o.ok = false; return; }  // not "one entry per line": caller falls back
        a--; b--;
        if (a < 0 || a >= M || b < 0 || b >= N) { o.range_error = true; o.bad_a = a + 1; o.bad_b = b + 1; return; }
        o.r.push_back((int)a); o.c.push_back((int)b); o.v.push_back((tilespmv::val_t)v);
    }
}

std::string lower(std::string s) { for (auto &c : s) c = (char)tolower((unsigned char)c); return s; }

}  // namespace

extern "C" int mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric, MAT_PTR_TYPE **csrRowPtr,
                             int **csrColIdx, MAT_VAL_TYPE **csrVal, char *filename)
{
    using namespace tilespmv;
    const int fd = open(filename, O_RDONLY);
    if (fd < 0) return -1;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); return sb.st_size == 0 ? -2 : -1; }
    const size_t got = (size_t)sb.st_size;
    void *map = mmap(nullptr, got, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return -1;
    (void)madvise(map, got, MADV_SEQUENTIAL);
    struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, got};
    const char *text = (const char *)map;
    Cursor c{text, text + got};

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

    struct Free { void *p; ~Free() { free(p); } };
    int *ri = (int *)malloc(sizeof(int) * (size_t)std::max(NZ, 1L)), *ci = (int *)malloc(sizeof(int) * (size_t)std::max(NZ, 1L));
    val_t *vv = (val_t *)malloc(sizeof(val_t) * (size_t)std::max(NZ, 1L));
    Free free_ri{ri}, free_ci{ci}, free_vv{vv};
    if (!ri || !ci || !vv) { fprintf(stderr, "mmio_allinone: out of host memory for %ld entries\n", NZ); return -4; }
    int *count = zalloc<int>((size_t)M + 1);
    const int kind = field == "real" ? 0 : field == "complex" ? 1 : field == "integer" ? 2 : 3;

    // ---- fast path: newline-aligned chunks parsed by all host threads, one entry per line
    bool parsed = false;
    {
        const char *body = c.p, *end = c.end;
        const size_t bytes = (size_t)(end - body);
        const int nchunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads() * 4, bytes / (1u << 20)));
        std::vector<const char *> cut((size_t)nchunk + 1);
        cut[0] = body; cut[(size_t)nchunk] = end;
        for (int k = 1; k < nchunk; k++) {
            const char *q = body + bytes / (size_t)nchunk * (size_t)k;
            if (q < cut[(size_t)k - 1]) q = cut[(size_t)k - 1];
            while (q < end && *q != '\n') q++;
            cut[(size_t)k] = q < end ? q + 1 : end;
        }
        std::vector<ChunkOut> outs((size_t)nchunk);
        parallel_chunks(nchunk, 1, [&](int64_t b0, int64_t b1, int) {
            for (int64_t k = b0; k < b1; k++) {
                outs[(size_t)k].r.reserve((size_t)(cut[(size_t)k + 1] - cut[(size_t)k]) / 12);
                parse_chunk(cut[(size_t)k], cut[(size_t)k + 1], kind, M, N, outs[(size_t)k]);
            }
        });
        bool ok = true;
        size_t total_entries = 0;
        for (auto &o : outs) {
            if (o.range_error) {
                fprintf(stderr, "mmio_allinone: entry (%ld,%ld) outside %ld x %ld in %s\n", o.bad_a, o.bad_b, M, N, filename);
                free(count);
                return -4;
            }
            ok = ok && o.ok;
            total_entries += o.r.size();
        }
        if (ok && total_entries >= (size_t)NZ) {  // extra lines beyond the declared count are ignored, like the reference's loop
            std::vector<size_t> at((size_t)nchunk + 1, 0);
            for (int k = 0; k < nchunk; k++) at[(size_t)k + 1] = std::min((size_t)NZ, at[(size_t)k] + outs[(size_t)k].r.size());
            parallel_chunks(nchunk, 1, [&](int64_t b0, int64_t b1, int) {
                for (int64_t k = b0; k < b1; k++) {
                    ChunkOut &o = outs[(size_t)k];
                    const size_t take = at[(size_t)k + 1] - at[(size_t)k];
                    if (take) {
                        memcpy(ri + at[(size_t)k], o.r.data(), take * sizeof(int));
                        memcpy(ci + at[(size_t)k], o.c.data(), take * sizeof(int));
                        memcpy(vv + at[(size_t)k], o.v.data(), take * sizeof(val_t));
                    }
                    std::vector<int>().swap(o.r); std::vector<int>().swap(o.c); std::vector<val_t>().swap(o.v);
                }
            });
            for (long i = 0; i < NZ; i++) count[ri[(size_t)i]]++;
            parsed = true;
        }
    }
    for (long i = 0; !parsed && i < NZ; i++) {
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

// ------------------------------------------------------------------------------------------------
// Writer (the reference's counterpart is mm_write_mtx_crd, src/mmio.h:605-645: one fprintf per entry): a general coordinate
// file in CSR (row-major) order, "real" with shortest-exact values ("%.17g", integers as integers) or "pattern".  Rows are
// formatted by all host threads into per-chunk buffers that are written out in order, so that a multi-GB file (the
// nlpkkt160 stand-in is ~4 GB of text) takes seconds.  Returns 0, -1 (cannot open), -3 (short write).
// ------------------------------------------------------------------------------------------------
namespace {

inline char *put_uint(char *p, unsigned long long v)
{
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

}  // namespace

extern "C" int tilespmv_mtx_write(const char *path, int m, int n, MAT_PTR_TYPE nnz, const MAT_PTR_TYPE *rowptr, const int *colidx,
                                  const MAT_VAL_TYPE *val /* NULL: pattern */)
{
    using namespace tilespmv;
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    bool ok = fprintf(f, "%%%%MatrixMarket matrix coordinate %s general\n%d %d %d\n", val ? "real" : "pattern", m, n, (int)nnz) > 0;
    const int64_t rows_per = 1 << 16;
    const int64_t nblk = ((int64_t)m + rows_per - 1) / rows_per;
    const int wave = std::max(1, host_threads());   // blocks formatted together, then written in order
    std::vector<std::vector<char>> buf((size_t)wave);
    for (int64_t b0 = 0; b0 < nblk && ok; b0 += wave) {
        const int64_t b1 = std::min(nblk, b0 + wave);
        parallel_chunks(b1 - b0, 1, [&](int64_t k0, int64_t k1, int) {
            for (int64_t k = k0; k < k1; k++) {
                const int64_t r0 = (b0 + k) * rows_per, r1 = std::min<int64_t>(m, r0 + rows_per);
                std::vector<char> &o = buf[(size_t)k];
                o.resize((size_t)(rowptr[r1] - rowptr[r0]) * 48 + 64);
                char *p = o.data();
                for (int64_t r = r0; r < r1; r++)
                    for (int j = rowptr[r]; j < rowptr[r + 1]; j++) {
                        p = put_uint(p, (unsigned long long)r + 1); *p++ = ' ';
                        p = put_uint(p, (unsigned long long)colidx[j] + 1);
                        if (val) {
                            *p++ = ' ';
                            const double v = (double)val[j];
                            // integer fast path only for finite values in range (the cast of Inf / NaN / 1e300 is undefined behaviour) and not for -0.0 (its sign would be lost)
                            if (std::isfinite(v) && v > -1e15 && v < 1e15 && v == (double)(long long)v && !(v == 0.0 && std::signbit(v))) {
                                long long iv = (long long)v;
                                if (iv < 0) { *p++ = '-'; iv = -iv; }
                                p = put_uint(p, (unsigned long long)iv);
                            } else p += snprintf(p, 32, "%.17g", v);
                        }
                        *p++ = '\n';
                    }
                o.resize((size_t)(p - o.data()));
            }
        });
        for (int64_t k = 0; k < b1 - b0 && ok; k++) ok = buf[(size_t)k].empty() || fwrite(buf[(size_t)k].data(), 1, buf[(size_t)k].size(), f) == buf[(size_t)k].size();
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : -3;
}
