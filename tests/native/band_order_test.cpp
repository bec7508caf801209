// C entry points over femcy_amd/csrc/band_order.hpp for tests/test_band_order_cpu.py (no HIP, no library state)
#include "../../femcy_amd/csrc/band_order.hpp"

extern "C" {

// rank[nn], node_at[nn] out; returns the half band width in nodes
int32_t bandtest_rcm(int32_t nn, int32_t ne, int32_t npe, const int32_t* elems, int32_t* rank, int32_t* node_at) {
    const femcy::BandOrder o = femcy::band_order_rcm(nn, ne, npe, elems);
    for (int32_t i = 0; i < nn; ++i) {
        rank[i] = o.rank[i];
        node_at[i] = o.node_at[i];
    }
    return o.half_band_nodes;
}

// A: lower band by columns (n * (bw + 1)), overwritten by L; sgn[n]; x[n] right-hand side in, solution out.
// returns 0 or 1 + the first zero / NaN pivot; *negative = number of negative pivots
int64_t bandtest_factor_solve(int64_t n, int64_t bw, double* A, double* sgn, double* x, int64_t* negative) {
    const int64_t bad = femcy::band_factor_host(n, bw, A, sgn, negative);
    if (!bad) femcy::band_solve_host(n, bw, A, sgn, x);
    return bad;
}
}
