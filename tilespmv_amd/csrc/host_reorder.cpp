// host_reorder.cpp — bandwidth-reducing renumbering for callers that STAY in the permuted numbering (a solver: permute x once at entry, run K products, un-permute y once
// at exit).  No reference counterpart (SURVEY S8 f4 "none in reference"): the reference multiplies the matrix in the numbering of its file (src/main.cu:59-110).
// Why: on meshes whose nodes were numbered in shuffled windows the tiles are ragged and an x gather touches several lines; reverse Cuthill-McKee on the symmetrised pattern
// brings the nonzeros back to a band — measured with scipy's ordering in round 5 (profiles/r05_rcm_probe.txt): kernel + 16-25 % (tet150s512 0.674 -> 0.783 of the roofline,
// tri2200s4096 0.713 -> 0.889), taken back when x and y are permuted around ONE product.  Host code, O(nnz log d): runs once, at plan creation.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "host_util.h"

using namespace tilespmv;

namespace {

// adjacency of the symmetrised pattern A + A^T without the diagonal, duplicates removed: ptr[n + 1], adj[ptr[n]]
void symmetrise(int n, const MAT_PTR_TYPE *rowptr, const int *colidx, std::vector<long long> &ptr, std::vector<int> &adj)
{
    std::vector<long long> cnt((size_t)n + 1, 0);
    for (int i = 0; i < n; i++)
        for (MAT_PTR_TYPE k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const int j = colidx[k];
            if (j != i && j >= 0 && j < n) { cnt[(size_t)i + 1]++; cnt[(size_t)j + 1]++; }
        }
    for (int i = 0; i < n; i++) cnt[(size_t)i + 1] += cnt[(size_t)i];
    std::vector<int> raw((size_t)cnt[(size_t)n]);
    std::vector<long long> at(cnt.begin(), cnt.end() - 1);
    for (int i = 0; i < n; i++)
        for (MAT_PTR_TYPE k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const int j = colidx[k];
            if (j != i && j >= 0 && j < n) { raw[(size_t)at[(size_t)i]++] = j; raw[(size_t)at[(size_t)j]++] = i; }
        }
    // sort + unique every list (threads over row ranges), then compact
    std::vector<int> len((size_t)n, 0);
    parallel_chunks((int64_t)n, 4096, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++) {
            int *lo = raw.data() + cnt[(size_t)i], *hi = raw.data() + cnt[(size_t)i + 1];
            std::sort(lo, hi);
            len[(size_t)i] = (int)(std::unique(lo, hi) - lo);
        }
    });
    ptr.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) ptr[(size_t)i + 1] = ptr[(size_t)i] + len[(size_t)i];
    adj.resize((size_t)ptr[(size_t)n]);
    parallel_chunks((int64_t)n, 4096, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; i++)
            if (len[(size_t)i]) memcpy(adj.data() + ptr[(size_t)i], raw.data() + cnt[(size_t)i], (size_t)len[(size_t)i] * sizeof(int));
    });
}

// Cuthill-McKee order of a graph (adjacency lists sorted, no self loops), REVERSED: order[new] = old.  Every connected component starts from a pseudo-peripheral node
// (George-Liu: repeat breadth-first searches from a node of the last level with the smallest degree while the depth grows), levels are filled in the order their parents were
// numbered, neighbours by ascending degree (ties: by number, so the result is deterministic).
void rcm_order(int n, const std::vector<long long> &ptr, const std::vector<int> &adj, std::vector<int> &out)
{
    auto deg = [&](int v) { return (int)(ptr[(size_t)v + 1] - ptr[(size_t)v]); };
    std::vector<int> order; order.reserve((size_t)n);
    std::vector<long long> mark((size_t)n, -1); // reached by the search with this stamp (pseudo-peripheral searches)
    std::vector<char> numbered((size_t)n, 0);
    std::vector<int> queue((size_t)std::max(n, 1)), level_of((size_t)n, 0), nb;
    long long stamp = 0;
    // breadth-first search from s over nodes that are not numbered yet; returns the depth, `last` = the node of the last level with the smallest degree
    auto bfs_depth = [&](int s, int &last) {
        int head = 0, tail = 0, depth = 0;
        queue[(size_t)tail++] = s; mark[(size_t)s] = stamp; level_of[(size_t)s] = 0;
        last = s;
        while (head < tail) {
            const int v = queue[(size_t)head++];
            const int lv = level_of[(size_t)v];
            if (lv > depth) { depth = lv; last = v; }
            else if (lv == depth && (deg(v) < deg(last) || (deg(v) == deg(last) && v < last))) last = v;
            for (long long k = ptr[(size_t)v]; k < ptr[(size_t)v + 1]; k++) {
                const int u = adj[(size_t)k];
                if (mark[(size_t)u] != stamp && !numbered[(size_t)u]) { mark[(size_t)u] = stamp; level_of[(size_t)u] = lv + 1; queue[(size_t)tail++] = u; }
            }
        }
        return depth;
    };
    // candidate roots: nodes by ascending degree (a component is entered at its node of smallest degree)
    std::vector<int> by_deg((size_t)n);
    std::iota(by_deg.begin(), by_deg.end(), 0);
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg(a) < deg(b); });
    for (int cand : by_deg) {
        if (numbered[(size_t)cand]) continue;
        int root = cand, last = cand;
        stamp++;
        int depth = bfs_depth(root, last);
        for (int it = 0; it < 8 && last != root; it++) {   // pseudo-peripheral node
            stamp++;
            int last2 = last;
            const int d2 = bfs_depth(last, last2);
            if (d2 <= depth) break;
            root = last; depth = d2; last = last2;
        }
        // Cuthill-McKee numbering of the component from `root`
        const size_t first = order.size();
        order.push_back(root); numbered[(size_t)root] = 1;
        for (size_t h = first; h < order.size(); h++) {
            const int v = order[h];
            nb.clear();
            for (long long k = ptr[(size_t)v]; k < ptr[(size_t)v + 1]; k++) { const int u = adj[(size_t)k]; if (!numbered[(size_t)u]) { numbered[(size_t)u] = 1; nb.push_back(u); } }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { const int da = deg(a), db = deg(b); return da != db ? da < db : a < b; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    out.assign(order.rbegin(), order.rend());
}

// Supervariables: nodes with the same CLOSED neighbourhood (adj(v) + v) — the unknowns of one mesh node in a multi-dof discretisation (3 displacements, 6 shell dofs) —
// are indistinguishable to any ordering and must stay together, or a tile's 16 rows mix the unknowns of many nodes.  super[v] = id of v's class, ids in order of each
// class's smallest member; returns the number of classes.  Candidates are found by a commutative hash of the closed neighbourhood and confirmed by comparing the lists.
int supervariables(int n, const std::vector<long long> &ptr, const std::vector<int> &adj, std::vector<int> &super)
{
    auto mix = [](unsigned long long x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
    std::vector<unsigned long long> h((size_t)n);
    parallel_chunks((int64_t)n, 4096, [&](int64_t b, int64_t e, int) {
        for (int64_t v = b; v < e; v++) {
            unsigned long long acc = mix((unsigned long long)v);
            for (long long k = ptr[(size_t)v]; k < ptr[(size_t)v + 1]; k++) acc += mix((unsigned long long)adj[(size_t)k]);
            h[(size_t)v] = acc;
        }
    });
    // u, v (u != v) have the same closed neighbourhood: same degree, adjacent, and the lists agree once u is taken out of adj(v) and v out of adj(u)
    auto same = [&](int u, int v) {
        const long long du = ptr[(size_t)u + 1] - ptr[(size_t)u], dv = ptr[(size_t)v + 1] - ptr[(size_t)v];
        if (du != dv) return false;
        const int *a = adj.data() + ptr[(size_t)u], *b = adj.data() + ptr[(size_t)v];
        if (!std::binary_search(a, a + du, v)) return false;
        long long i = 0, j = 0;
        while (i < du || j < dv) {
            if (i < du && a[i] == v) { i++; continue; }
            if (j < dv && b[j] == u) { j++; continue; }
            if (i >= du || j >= dv || a[i] != b[j]) return false;
            i++; j++;
        }
        return true;
    };
    std::vector<int> by_hash((size_t)n);
    std::iota(by_hash.begin(), by_hash.end(), 0);
    std::sort(by_hash.begin(), by_hash.end(), [&](int a, int b) { return h[(size_t)a] != h[(size_t)b] ? h[(size_t)a] < h[(size_t)b] : a < b; });
    std::vector<int> rep((size_t)n);   // smallest member of v's class
    std::iota(rep.begin(), rep.end(), 0);
    std::vector<int> reps;
    for (size_t g0 = 0; g0 < (size_t)n;) {
        size_t g1 = g0 + 1;
        while (g1 < (size_t)n && h[(size_t)by_hash[g1]] == h[(size_t)by_hash[g0]]) g1++;
        if (g1 - g0 > 1) {   // (members ascend inside a hash group: the first member of a class is its smallest)
            reps.clear();
            for (size_t q = g0; q < g1; q++) {
                const int v = by_hash[q];
                bool found = false;
                for (int r : reps) if (same(r, v)) { rep[(size_t)v] = r; found = true; break; }
                if (!found) reps.push_back(v);
            }
        }
        g0 = g1;
    }
    super.assign((size_t)n, -1);
    int ns = 0;
    for (int v = 0; v < n; v++) if (rep[(size_t)v] == v) super[(size_t)v] = ns++;
    for (int v = 0; v < n; v++) super[(size_t)v] = super[(size_t)rep[(size_t)v]];
    return ns;
}

}  // namespace

extern "C" {

// Reverse Cuthill-McKee on the symmetrised pattern of the leading n x n block of a CSR matrix (columns >= n — a halo — are ignored): perm[new] = old.
// Where the pattern has supervariables (classes of >= 1.2 nodes on average: multi-dof meshes) the ordering runs on the quotient graph and the members of a class stay
// together, in ascending original order; otherwise on the graph itself.  Deterministic.  Returns 0, or -1 on bad arguments / allocation failure.
int tilespmv_reorder_rcm(int n, const MAT_PTR_TYPE *rowptr, const int *colidx, int *perm)
{
    if (n < 0 || (n > 0 && (!rowptr || !perm))) return -1;
    if (n == 0) return 0;
    try {
        std::vector<long long> ptr; std::vector<int> adj;
        symmetrise(n, rowptr, colidx, ptr, adj);
        std::vector<int> super, order;
        const int ns = supervariables(n, ptr, adj, super);
        if ((long long)ns * 6 <= (long long)n * 5) {
            // quotient graph: the class of every neighbour of a class's first member (all members have the same neighbours outside the class), without the class itself
            std::vector<int> first((size_t)ns, -1);
            for (int v = n - 1; v >= 0; v--) first[(size_t)super[(size_t)v]] = v;
            std::vector<long long> qptr((size_t)ns + 1, 0);
            std::vector<std::vector<int>> lists((size_t)ns);
            parallel_chunks((int64_t)ns, 1024, [&](int64_t b, int64_t e, int) {
                for (int64_t q = b; q < e; q++) {
                    const int r = first[(size_t)q];
                    std::vector<int> &L = lists[(size_t)q];
                    for (long long k = ptr[(size_t)r]; k < ptr[(size_t)r + 1]; k++) { const int c = super[(size_t)adj[(size_t)k]]; if (c != (int)q) L.push_back(c); }
                    std::sort(L.begin(), L.end());
                    L.erase(std::unique(L.begin(), L.end()), L.end());
                }
            });
            for (int q = 0; q < ns; q++) qptr[(size_t)q + 1] = qptr[(size_t)q] + (long long)lists[(size_t)q].size();
            std::vector<int> qadj((size_t)qptr[(size_t)ns]);
            for (int q = 0; q < ns; q++) if (!lists[(size_t)q].empty()) memcpy(qadj.data() + qptr[(size_t)q], lists[(size_t)q].data(), lists[(size_t)q].size() * sizeof(int));
            std::vector<std::vector<int>>().swap(lists);
            std::vector<int> qorder;
            rcm_order(ns, qptr, qadj, qorder);
            // expand: the members of each class in ascending original order
            std::vector<long long> mptr((size_t)ns + 1, 0);
            for (int v = 0; v < n; v++) mptr[(size_t)super[(size_t)v] + 1]++;
            for (int q = 0; q < ns; q++) mptr[(size_t)q + 1] += mptr[(size_t)q];
            std::vector<int> members((size_t)n);
            { std::vector<long long> at(mptr.begin(), mptr.end() - 1); for (int v = 0; v < n; v++) members[(size_t)at[(size_t)super[(size_t)v]]++] = v; }
            order.reserve((size_t)n);
            for (int q : qorder) for (long long k = mptr[(size_t)q]; k < mptr[(size_t)q + 1]; k++) order.push_back(members[(size_t)k]);
        } else rcm_order(n, ptr, adj, order);
        if ((int)order.size() != n) return -1;
        for (int i = 0; i < n; i++) perm[i] = order[(size_t)i];
    } catch (const std::bad_alloc &) { return -1; }
    return 0;
}

// B = P A P^T for perm[new] = old: row i of B is row perm[i] of A, a column j < n becomes inverse[j], columns >= n (the halo part of a rank's [own | halo] index
// space) stay; the entries of every row of B are in ASCENDING column order (stable for repeated columns) — not in A's order mapped through the permutation: the reference's
// dense-row / dense-col tiles take the columns of a row to ascend (src/csr2tile.h:586,600-605 against src/tilespmv_cpu.h:246,262), and a renumbering scrambles them.
// val / out_val may be NULL (pattern only).  out_rowptr[n + 1], out_colidx / out_val[nnz].  Returns 0 / -1.
int tilespmv_csr_permute(int n, const MAT_PTR_TYPE *rowptr, const int *colidx, const MAT_VAL_TYPE *val, const int *perm,
                         MAT_PTR_TYPE *out_rowptr, int *out_colidx, MAT_VAL_TYPE *out_val)
{
    if (n < 0 || (n > 0 && (!rowptr || !perm || !out_rowptr))) return -1;
    std::vector<int> inv((size_t)std::max(n, 1), -1);
    for (int i = 0; i < n; i++) { if (perm[i] < 0 || perm[i] >= n || inv[(size_t)perm[i]] != -1) return -1; inv[(size_t)perm[i]] = i; }
    out_rowptr[0] = 0;
    for (int i = 0; i < n; i++) out_rowptr[i + 1] = out_rowptr[i] + (rowptr[perm[i] + 1] - rowptr[perm[i]]);
    parallel_chunks((int64_t)n, 2048, [&](int64_t b, int64_t e, int) {
        std::vector<unsigned long long> key;
        for (int64_t i = b; i < e; i++) {
            const MAT_PTR_TYPE s = rowptr[perm[i]], len = rowptr[perm[i] + 1] - s, d = out_rowptr[i];
            key.resize((size_t)len);
            for (MAT_PTR_TYPE k = 0; k < len; k++) {
                const int j = colidx[s + k];
                key[(size_t)k] = ((unsigned long long)(unsigned)((j >= 0 && j < n) ? inv[(size_t)j] : j) << 32) | (unsigned)k;   // (new column, position in A's row): sorts stably
            }
            std::sort(key.begin(), key.end());
            for (MAT_PTR_TYPE k = 0; k < len; k++) {
                out_colidx[d + k] = (int)(key[(size_t)k] >> 32);
                if (val && out_val) out_val[d + k] = val[s + (MAT_PTR_TYPE)(key[(size_t)k] & 0xFFFFFFFFull)];
            }
        }
    });
    return 0;
}

// Half bandwidth max |i - j| over the leading n x n block (a plan fact for reports)
long long tilespmv_csr_bandwidth(int n, const MAT_PTR_TYPE *rowptr, const int *colidx)
{
    long long bw = 0;
    for (int i = 0; i < n; i++)
        for (MAT_PTR_TYPE k = rowptr[i]; k < rowptr[i + 1]; k++) { const int j = colidx[k]; if (j >= 0 && j < n) bw = std::max(bw, (long long)std::abs(i - j)); }
    return bw;
}

}  // extern "C"
