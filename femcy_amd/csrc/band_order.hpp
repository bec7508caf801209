// Direct solve of small systems, host part shared by both libraries (no HIP here): the row order that makes K a narrow
// band, and -- for libfemcy_cpu.so -- the band Cholesky itself.
//
// stiffnessMtrx.py:219-251 (`solve_by_scipy`): below 1e5 DOF the reference hands K to scipy's sparse direct solver.
// Here the nodes are renumbered by reverse Cuthill-McKee (a structured or graded FEM mesh of that size becomes a band of
// a few hundred to a few thousand DOF), K is copied into lower band storage and factored K = L S L^T, S = diag(+-1),
// without pivoting: plain Cholesky for the positive definite K of a sound configuration (after the Dirichlet
// treatment, :279-341: unit rows / columns), and a factorisation all the same when a diverging Newton iterate has
// inverted elements and K is indefinite.  The residual of the solution is checked and refined against K itself; a zero
// pivot or a residual that stays large is reported (the caller treats it like a numerical breakdown).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <cmath>
#include <vector>

namespace femcy {

// residual policy of the direct solve (both libraries): a factorisation of a positive definite K leaves 1e-14 ... 1e-10
// depending on |K| |x| / |b| (the sparse LU the reference calls leaves the same); refine while max|b - K x| > REFINE_ABOVE max|b| and a step still
// halves it, at most MAX_REFINE times; a solution is returned only if it ends at or below ACCEPT
// (round 5: REFINE_ABOVE 1e-12 -> 1e-10.  On the wide bands of cube-like 3-D meshes the unrefined residual is 1.4e-12 ...
// 6.5e-12 -- what a sparse LU leaves -- and 1e-12 bought it a whole extra forward + backward sweep, 8 us per panel: 5.4 ->
// 3.8, 29.1 -> 22.3, 128.5 -> 104.5 ms per solve; the 49 decks end to end are unchanged, worst 2.0e-10 rel. L2 at
// nu = 0.4999, where the refinement still runs: profiles/r05_direct_refine_threshold.txt)
#ifndef FEMCY_DIRECT_REFINE_ABOVE
#define FEMCY_DIRECT_REFINE_ABOVE 1e-10
#endif
constexpr double DIRECT_REFINE_ABOVE = FEMCY_DIRECT_REFINE_ABOVE, DIRECT_ACCEPT = 1e-8;
constexpr int DIRECT_MAX_REFINE = 2;

struct BandOrder {
    std::vector<int32_t> rank;      // node -> position in the band order
    std::vector<int32_t> node_at;   // position -> node
    int32_t half_band_nodes = 0;    // max |rank[a] - rank[b]| over coupled nodes
};

// Reverse Cuthill-McKee on the node graph of the mesh (two nodes are coupled when an element holds both).  Every
// connected component starts from a pseudo-peripheral node (George & Liu: repeat the breadth-first search from a node
// of least degree in the last level until the depth stops growing); inside a level the nodes follow their parents,
// ties by ascending degree.  Deterministic for a given connectivity.
inline BandOrder band_order_rcm(int32_t nn, int32_t ne, int32_t npe, const int32_t* elems) {
    // node -> incident elements
    std::vector<int64_t> eptr((size_t)nn + 1, 0);
    for (int64_t i = 0; i < (int64_t)ne * npe; ++i) ++eptr[elems[i] + 1];
    for (int32_t a = 0; a < nn; ++a) eptr[a + 1] += eptr[a];
    std::vector<int32_t> eidx((size_t)ne * npe);
    {
        std::vector<int64_t> fill(eptr.begin(), eptr.end() - 1);
        for (int32_t e = 0; e < ne; ++e)
            for (int32_t k = 0; k < npe; ++k) eidx[fill[elems[(int64_t)e * npe + k]]++] = e;
    }
    // adjacency (distinct neighbours, the node itself excluded)
    std::vector<int64_t> aptr((size_t)nn + 1, 0);
    std::vector<int32_t> adj;
    {
        std::vector<int32_t> mark((size_t)nn, -1);
        adj.reserve((size_t)ne * npe * 2);
        for (int32_t a = 0; a < nn; ++a) {
            mark[a] = a;
            for (int64_t q = eptr[a]; q < eptr[a + 1]; ++q) {
                const int32_t* en = elems + (int64_t)eidx[q] * npe;
                for (int32_t k = 0; k < npe; ++k) {
                    const int32_t b = en[k];
                    if (mark[b] != a) {
                        mark[b] = a;
                        adj.push_back(b);
                    }
                }
            }
            aptr[a + 1] = (int64_t)adj.size();
        }
    }
    auto degree = [&](int32_t a) { return (int32_t)(aptr[a + 1] - aptr[a]); };
    for (int32_t a = 0; a < nn; ++a)   // neighbours by ascending degree (ties by number): the order they are queued in
        std::sort(adj.begin() + aptr[a], adj.begin() + aptr[a + 1], [&](int32_t x, int32_t y) {
            const int32_t dx = degree(x), dy = degree(y);
            return dx != dy ? dx < dy : x < y;
        });

    std::vector<int32_t> order;      // Cuthill-McKee order (reversed at the end)
    order.reserve((size_t)nn);
    std::vector<int32_t> level((size_t)nn, -1), stamp((size_t)nn, -1), queue;
    std::vector<char> done((size_t)nn, 0);
    int32_t bfs_id = 0;
    // breadth-first search from `root` over nodes not yet ordered: fills queue, returns depth and a node of least
    // degree in the last level
    auto bfs = [&](int32_t root, int32_t& last_min) {
        queue.clear();
        queue.push_back(root);
        stamp[root] = bfs_id;
        level[root] = 0;
        for (size_t h = 0; h < queue.size(); ++h) {
            const int32_t a = queue[h];
            for (int64_t q = aptr[a]; q < aptr[a + 1]; ++q) {
                const int32_t b = adj[q];
                if (!done[b] && stamp[b] != bfs_id) {
                    stamp[b] = bfs_id;
                    level[b] = level[a] + 1;
                    queue.push_back(b);
                }
            }
        }
        ++bfs_id;
        const int32_t depth = level[queue.back()];
        last_min = queue.back();
        for (size_t h = queue.size(); h-- > 0 && level[queue[h]] == depth;)
            if (degree(queue[h]) < degree(last_min) || (degree(queue[h]) == degree(last_min) && queue[h] < last_min))
                last_min = queue[h];
        return depth;
    };
    for (int32_t seed = 0; seed < nn; ++seed) {
        if (done[seed]) continue;
        int32_t root = seed, far = seed;
        int32_t depth = bfs(root, far);
        for (int pass = 0; pass < 8; ++pass) {           // towards a pseudo-peripheral node
            int32_t far2 = far;
            const int32_t d2 = bfs(far, far2);
            if (d2 <= depth) break;
            root = far;
            far = far2;
            depth = d2;
            // (the search from `root` is the one just made)
        }
        int32_t dummy;
        bfs(root, dummy);
        for (int32_t a : queue) {
            done[a] = 1;
            order.push_back(a);
        }
    }
    BandOrder o;
    o.rank.assign((size_t)nn, 0);
    o.node_at.assign((size_t)nn, 0);
    for (int32_t i = 0; i < nn; ++i) {
        const int32_t a = order[(size_t)nn - 1 - i];
        o.node_at[i] = a;
        o.rank[a] = i;
    }
    int32_t hb = 0;
    for (int32_t a = 0; a < nn; ++a)
        for (int64_t q = aptr[a]; q < aptr[a + 1]; ++q) hb = std::max(hb, std::abs(o.rank[a] - o.rank[adj[q]]));
    o.half_band_nodes = hb;
    return o;
}

// Host band factorisation (libfemcy_cpu.so): K = L S L^T, L lower triangular with a positive diagonal, S = diag(+-1) --
// Cholesky when K is positive definite (every sign +1), and still a factorisation when a diverging Newton iterate has
// made K indefinite (the reference's LU solves those systems too, and the increment driver's path depends on what
// comes back).  No pivoting: the caller checks the residual and refines.
// Lower band by columns: entry (i, j), j <= i <= j + bw, at A[j * (bw + 1) + (i - j)].  Right-looking: column j is
// scaled, then every later column k of its window takes its rank-1 share -- contiguous in both operands, spread over
// the threads of the parallel region.  Returns 0, or 1 + the first column whose pivot is zero or not a number;
// *negative = number of negative pivots.
inline int64_t band_factor_host(int64_t n, int64_t bw, double* A, double* sgn, int64_t* negative) {
    const int64_t w = bw + 1;
    int64_t bad = 0, neg = 0;
#pragma omp parallel
    {
        for (int64_t j = 0; j < n; ++j) {
            const int64_t iend = std::min(n, j + bw + 1);
            double* cj = A + j * w - j;                        // cj[i] = entry (i, j)
#pragma omp single
            {
                const double d = cj[j], a = std::fabs(d);
                const double sj = d < 0.0 ? -1.0 : 1.0;
                if (!(a > 0.0) && bad == 0) bad = j + 1;
                if (d < 0.0) ++neg;
                const double piv = a > 0.0 ? std::sqrt(a) : 1.0;
                cj[j] = piv;
                sgn[j] = sj;
                const double inv = sj / piv;
                for (int64_t i = j + 1; i < iend; ++i) cj[i] *= inv;
            }   // implicit barrier
            const double sj = sgn[j];
#pragma omp for schedule(static)
            for (int64_t k = j + 1; k < iend; ++k) {
                const double lkj = cj[k] * sj;
                if (lkj == 0.0) continue;
                double* ck = A + k * w - k;
                for (int64_t i = k; i < iend; ++i) ck[i] -= cj[i] * lkj;
            }   // implicit barrier
        }
    }
    *negative = neg;
    return bad;
}

// L z = b, w = S z, L^T x = w, in place
inline void band_solve_host(int64_t n, int64_t bw, const double* L, const double* sgn, double* x) {
    const int64_t w = bw + 1;
    for (int64_t j = 0; j < n; ++j) {
        const double* cj = L + j * w - j;
        const int64_t iend = std::min(n, j + bw + 1);
        const double xj = x[j] / cj[j];
        x[j] = xj;
        for (int64_t i = j + 1; i < iend; ++i) x[i] -= cj[i] * xj;
    }
    for (int64_t j = n; j-- > 0;) {
        const double* cj = L + j * w - j;
        const int64_t iend = std::min(n, j + bw + 1);
        double s = x[j] * sgn[j];
        for (int64_t i = j + 1; i < iend; ++i) s -= cj[i] * x[i];
        x[j] = s / cj[j];
    }
}

}  // namespace femcy
