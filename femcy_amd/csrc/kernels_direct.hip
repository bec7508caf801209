// Direct solve of small systems on the device: band factorisation K = L S L^T (Cholesky when K is positive definite).
//
//   solve_by_scipy   stiffnessMtrx.py:219-251   -> direct_solve (femcy_direct_solve)
//
// The reference hands systems below 1e5 DOF to scipy's sparse direct solver on the host.  Here the factorisation runs
// on the GPU, on a matrix that never leaves it:
//   * the nodes are renumbered by reverse Cuthill-McKee (band_order.hpp, once per pattern): K becomes a band of
//     bw = (max rank distance + 1) * dm - 1 sub-diagonals -- a few hundred DOF for the 2-D decks, a few thousand for a
//     3-D mesh of 3e4 nodes;
//   * the lower band is stored as TILES of 32 x 32 doubles (8 KB, row-major), column panel after column panel:
//     tile (p, t) = rows of block row p + t, columns of panel p, t = 0 .. T with T = ceil(bw / 32) -- a tile is what
//     one workgroup reads or writes with full 128-byte lines, and fill-in stays inside the band;
//   * K = L S L^T with L lower triangular (positive diagonal) and S = diag(+-1): Cholesky when K is positive definite,
//     and still a factorisation when a diverging Newton iterate has inverted elements and K is indefinite -- the
//     reference's LU returns a solution there too and the increment driver's path depends on it.  No pivoting (it
//     would leave the band); instead the residual b - K x is formed with K itself (the SpMV of the PCG) and the
//     solution refined while that pays, and a result is returned only if the residual ends small (band_order.hpp);
//   * right-looking factorisation, two launches per panel: k_band_panel (every workgroup factors the 32 x 32 diagonal
//     tile in the registers of its wave -- redundantly: 6 kflop against a grid-wide hand-over -- and solves ITS tile of
//     the panel against it in the same instruction stream),
//     k_band_update (one workgroup per pair of panel tiles: a 32 x 32 x 32 product subtracted from the tile it meets);
//   * the forward sweep of the first right-hand side rides inside k_band_panel (b as a 33rd column); the backward sweep
//     (and both sweeps of a refinement solve) take one launch per panel, column-oriented so that the T tiles of a panel
//     are independent workgroups: the 32 unknowns of the panel are solved in registers of one wave (lane = row, its row
//     of the factor in registers, v_readlane broadcast of the solved unknown), then every tile subtracts its share
//     from the rows it couples to.
// All launches sit on the context's stream in order; what returns to the host, in one synchronisation, is the pivot
// flags and two residual norms.  The arithmetic is f64 VALU in the panel and sweep kernels: what is waited for there is
// the chain of dependent one-wave launches (~20 us per panel of 32 unknowns), not bytes or flops.  The trailing update
// -- the one dense small GEMM of the repository -- runs on the f64 matrix cores from 8 tiles per panel on (round 5).
//
// State lives beside the context (a table keyed by the context's address), created on first use and dropped by
// femcy_ctx_destroy / a new pattern.
#include <mutex>
#include <unordered_map>
#include "band_order.hpp"
#include "ctx.hpp"

namespace femcy {

namespace {

constexpr int NB = 32;          // panel width = tile edge (DOF)
// the trailing update runs on the f64 matrix cores from this many tiles per panel on (k_band_update_mfma, variant 1: one
// tile pair per workgroup).  Measured (profiles/r05_direct_mfma_update.txt): 89 k-DOF cube 104.2 -> 85.6 ms, 27.8 k 22.3 ->
// 20.2, the C3D10 twist deck 3.90 -> 3.61, the dense CPS6 deck 17.80 -> 16.67; variant 2 (2 x 2 tile pairs per workgroup)
// is no faster than the VALU product (105.3 ms: a quarter of the workgroups, the same chain of target load -> stage ->
// barrier -> product -> store in each): the update is bound by that latency chain, not by tile traffic
constexpr int DIRECT_MFMA_MIN_TILES = 8, DIRECT_MFMA_VARIANT = 1;
// variant 3 (the update on two streams: the column the next panel needs first, the rest beside that panel) is NOT a default:
// measured slower at every size -- 89 k-DOF cube 86.1 -> 100.2 ms, 27.8 k 20.3 -> 28.0, dense CPS6 deck 16.75 -> 28.44
// (profiles/r05_direct_mfma_update.txt): two event hand-overs per panel between the streams cost more than the
// overlap of a 15 us update with a 10 us panel returns.  Kept as a tested knob.
constexpr int DIRECT_OVERLAP_MIN_TILES = 24, DIRECT_OVERLAP_VARIANT = -1;
constexpr int TS = NB * NB;     // doubles per tile

struct DirectState {
    int64_t pattern_serial = -1;
    int64_t max_bytes = (int64_t)48 << 30;
    int32_t half_band_nodes = 0, bw = 0, T = 0, P = 0;
    int64_t band_tiles = 0;
    int32_t* d_rank = nullptr;
    int32_t* d_node_at = nullptr;
    double* d_band = nullptr;
    double* d_dfac = nullptr;
    double* d_sgn = nullptr;      // [P * 32] the signs S
    double* d_invd = nullptr;     // [P * 32] 1 / l_kk
    double *d_wb = nullptr, *d_wy = nullptr, *d_wx = nullptr;
    double *d_res = nullptr, *d_Kx = nullptr;   // [n] caller's numbering: residual, K x
    double* d_norms = nullptr;    // max|res| (NaN if any entry is), max|b|
    int32_t* d_flag = nullptr;    // [0] 1 + first pivot that is zero / NaN, [1] negative pivots
    char* h_back = nullptr;       // pinned: 2 int32 + 2 doubles
    int update_variant = -1;      // FEMCY_TUNE_DIRECT_UPDATE: -1 auto, 0 VALU, 1 / 2 matrix cores (k_band_update_mfma / _mfma2),
                                  // 3 = 1 on two streams (the update beside the next panel)
    static constexpr int NEV = 16;
    hipStream_t s2 = nullptr;     // second stream + event ring of the two-stream schedule (created on first use)
    hipEvent_t ev[NEV] = {nullptr};
    BandOrder order;              // host copy of the band order of pattern `order_serial`
    int64_t order_serial = -1;
    void release() {
        for (void* q : {(void*)d_rank, (void*)d_node_at, (void*)d_band, (void*)d_dfac, (void*)d_sgn, (void*)d_invd, (void*)d_wb, (void*)d_wy,
                        (void*)d_wx, (void*)d_res, (void*)d_Kx, (void*)d_norms, (void*)d_flag})
            if (q) (void)hipFree(q);
        if (h_back) (void)hipHostFree(h_back);
        if (s2) {
            (void)hipStreamSynchronize(s2);
            (void)hipStreamDestroy(s2);
            for (hipEvent_t& e : ev) {
                if (e) (void)hipEventDestroy(e);
                e = nullptr;
            }
            s2 = nullptr;
        }
        d_rank = d_node_at = d_flag = nullptr;
        d_band = d_dfac = d_sgn = d_invd = d_wb = d_wy = d_wx = d_res = d_Kx = d_norms = nullptr;
        h_back = nullptr;
        pattern_serial = -1;
    }
};

std::mutex g_mu;
std::unordered_map<const Ctx*, DirectState> g_states;

DirectState& state_of(const Ctx* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_states[c];   // references into an unordered_map stay valid across insertions
}

// ---------------------------------------------------------------------------------------------------------------
// K (blocked SELL, caller's numbering) -> lower band tiles (band order).  Thread = (slice, slot j, lane).
template <int DM>
__global__ void __launch_bounds__(256) k_band_fill(int32_t nslices, int32_t maxL, const int32_t* __restrict__ slice_len,
                                                   const int64_t* __restrict__ slice_off,
                                                   const int32_t* __restrict__ node_of, const int32_t* __restrict__ rowlen,
                                                   const int32_t* __restrict__ bcol, const double* __restrict__ Kvals,
                                                   const int32_t* __restrict__ rank, int32_t T, double* __restrict__ band) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(t & 63);
    const int64_t q = t >> 6;
    const int32_t j = (int32_t)(q % maxL), s = (int32_t)(q / maxL);
    if (s >= nslices || j >= slice_len[s]) return;
    const int32_t a = node_of[(int64_t)s * SLICE + lane];
    if (a < 0 || j >= rowlen[a]) return;
    const int64_t row = slice_off[s] + j;
    const int32_t b = bcol[row * SLICE + lane];
    const int32_t ra = rank[a], rb = rank[b];
    if (ra < rb) return;                                          // the upper triangle is the mirror lane's
#pragma unroll
    for (int r = 0; r < DM; ++r)
#pragma unroll
        for (int cc = 0; cc < DM; ++cc) {
            const int32_t i = ra * DM + r, jj = rb * DM + cc;
            if (i < jj) continue;
            const int32_t pn = jj / NB, tt = i / NB - pn;
            band[((int64_t)pn * (T + 1) + tt) * TS + (i % NB) * NB + (jj % NB)] = Kvals[kv_index<DM>(row, r * DM + cc, lane)];
        }
}

// right-hand side into band order; the rows that pad the last panel are unit rows
__global__ void __launch_bounds__(256) k_band_gather(int64_t n, int32_t P, int32_t dm, int32_t T,
                                                     const int32_t* __restrict__ node_at, const double* __restrict__ b,
                                                     double* __restrict__ wb, double* __restrict__ band) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)P * NB) return;
    if (i < n) {
        wb[i] = b[(int64_t)node_at[i / dm] * dm + i % dm];
    } else {
        wb[i] = 0.0;
        if (band) band[((int64_t)(i / NB) * (T + 1)) * TS + (i % NB) * NB + (i % NB)] = 1.0;   // (not when refining)
    }
}

__global__ void __launch_bounds__(256) k_band_scatter(int64_t n, int32_t dm, const int32_t* __restrict__ node_at,
                                                      const double* __restrict__ wx, double* __restrict__ x, int add) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double* dst = x + (int64_t)node_at[i / dm] * dm + i % dm;
    *dst = add ? *dst + wx[i] : wx[i];
}

// res = b - K x, norms[0] = max|res| (NaN as soon as one entry is), norms[1] = max|b|.  One workgroup: n < 1e5 here.
__global__ void __launch_bounds__(1024) k_band_residual(int64_t n, const double* __restrict__ b, const double* __restrict__ Kx,
                                                        double* __restrict__ res, double* __restrict__ norms) {
    __shared__ double sr[1024], sb[1024];
    __shared__ int snan;
    if (threadIdx.x == 0) snan = 0;
    __syncthreads();
    double rm = 0.0, bm = 0.0;
    bool nan = false;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double r = b[i] - Kx[i];
        res[i] = r;
        nan |= r != r;
        rm = fmax(rm, fabs(r));
        bm = fmax(bm, fabs(b[i]));
    }
    if (nan) snan = 1;
    sr[threadIdx.x] = rm;
    sb[threadIdx.x] = bm;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sr[threadIdx.x] = fmax(sr[threadIdx.x], sr[threadIdx.x + o]);
            sb[threadIdx.x] = fmax(sb[threadIdx.x], sb[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        norms[0] = snan ? __builtin_nan("") : sr[0];
        norms[1] = sb[0];
    }
}

// value of lane `l` (uniform; a compile-time constant in the unrolled loops below) as a scalar
__device__ __forceinline__ double lane_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// sqrt(a) and 1 / sqrt(a) of a pivot, a > 0: the hardware's reciprocal-square-root estimate, two Newton steps, one
// correction each (2 ulp; the elimination's step-to-step chain waits for exactly this, and the library sqrt + division
// it replaces is twice as long -- the residual check downstream does not care about the last bit)
__device__ __forceinline__ void pivot_roots(double a, double& root, double& inv_root) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    double g = a * y;
    g = fma(fma(-g, g, a), 0.5 * y, g);                           // sqrt: g += (a - g^2) / (2 g)
    y = fma(fma(-g, y, 1.0), y, y);                               // 1 / g: y += y (1 - g y)
    root = g;
    inv_root = y;
}

// The elimination of k_band_panel, written as compile-time recursion over the step k and the column j so that every
// register index and every broadcast lane is a constant AND the stages of the next pivot's root chain can be placed
// between the updates of the current step.  The pivot of step k + 1 is final after the FIRST update of step k; its
// roots are a dozen dependent operations, and a wave issues in order: cut into stages and interleaved with the
// remaining updates (independent of them), the chain's latency passes in the shadow of work that is due anyway.
struct PivotRoots {                                               // sqrt(a) -> g, 1 / sqrt(a) -> y (see pivot_roots)
    double a, h, y, t, g, e;
};
constexpr int ROOT_STAGES = 12;
template <int S>
__device__ __forceinline__ void root_stage(PivotRoots& x) {
    if constexpr (S == 0) {
        x.y = __builtin_amdgcn_rsq(x.a);
        x.h = -0.5 * x.a;
    } else if constexpr (S == 1 || S == 4) {
        x.t = x.h * x.y;
    } else if constexpr (S == 2 || S == 5) {
        x.t = fma(x.t, x.y, 1.5);
    } else if constexpr (S == 3 || S == 6) {
        x.y = x.y * x.t;
    } else if constexpr (S == 7) {
        x.g = x.a * x.y;
    } else if constexpr (S == 8) {
        x.e = fma(-x.g, x.g, x.a);
        x.t = 0.5 * x.y;
    } else if constexpr (S == 9) {
        x.g = fma(x.e, x.t, x.g);
    } else if constexpr (S == 10) {
        x.e = fma(-x.g, x.y, 1.0);
    } else if constexpr (S == 11) {
        x.y = fma(x.e, x.y, x.y);
    }
}
struct PanelRow {                                                 // what a lane carries besides its 32 entries
    double sign, inv, bi, zi;
    int32_t bad_at, negative;
};
template <int K, int J>
__device__ __forceinline__ void panel_updates(double (&r)[NB], double rks, PivotRoots& nx) {
    if constexpr (J < NB) {
        if constexpr (J - (K + 2) < ROOT_STAGES) root_stage<J - (K + 2)>(nx);
        r[J] -= rks * lane_bcast(r[K], J);
        panel_updates<K, J + 1>(r, rks, nx);
    }
}
template <int K, int Q>
__device__ __forceinline__ void root_tail(PivotRoots& nx) {       // the late steps have fewer updates than stages
    if constexpr (Q < ROOT_STAGES) {
        if constexpr (Q >= NB - (K + 2)) root_stage<Q>(nx);
        root_tail<K, Q + 1>(nx);
    }
}
template <int K>
__device__ __forceinline__ void panel_step(double (&r)[NB], PanelRow& st, int lane, int32_t p, double dkk, double piv,
                                           double ipiv) {
    if constexpr (K < NB) {
        const bool bad = !(fabs(dkk) > 0.0);                      // zero or NaN
        const double sk = dkk < 0.0 ? -1.0 : 1.0;
        if (bad && st.bad_at == 0) st.bad_at = p * NB + K + 1;
        st.negative += dkk < 0.0 ? 1 : 0;
        if (lane == K) {
            st.sign = sk;
            st.inv = ipiv;
        }
        const double scaled = r[K] * (sk * ipiv);
        r[K] = (lane == K) ? piv : ((lane > K) ? scaled : r[K]);  // rows above K: the (unused) upper triangle stays
        const double rks = r[K] * sk;
        const double zk = lane_bcast(st.bi, K) * ipiv;
        double dnext = 0.0;
        PivotRoots nx;
        nx.a = 1.0;
        if constexpr (K + 1 < NB) {
            r[K + 1] -= rks * lane_bcast(r[K], K + 1);
            dnext = lane_bcast(r[K + 1], K + 1);
            nx.a = fabs(dnext) > 0.0 ? fabs(dnext) : 1.0;
        }
        panel_updates<K, K + 2>(r, rks, nx);
        root_tail<K, 0>(nx);
        st.zi = (lane == K) ? zk : st.zi;
        st.bi = (lane > K) ? st.bi - r[K] * zk : st.bi;
        panel_step<K + 1>(r, st, lane, p, dnext, nx.g, nx.y);
    }
}

// panel p: workgroup t = one wave.  Lanes 0..31 hold the rows of the diagonal tile, lanes 32..63 the rows of tile t of
// the panel (workgroup 0: copies of the diagonal rows, never stored) -- 32 doubles per lane, in registers.  The
// right-looking elimination of the diagonal tile, A = L S L^T, IS the triangular solve of the rows below it: at step k
// every row scales its entry of column k by s_k / l_kk and subtracts its multiple of column k from its later entries,
// L(j, k) broadcast from lane j -- one instruction stream for both halves of the wave, no LDS, no barrier (the first
// version, LDS tiles + three barriers per step, took 51 us per panel against 8: profiles/r04_direct_kernels.txt).
// Workgroup 0 stores L to dfac[p], the signs to sgn, 1 / l_kk to invd; workgroup t > 0 replaces its tile by A (S L^T)^-1.
// The right-hand side rides along as one more column: row i carries b_i, at step k z_k = b_k / l_kk goes to all and the
// rows below subtract their multiple -- the forward sweep of the first solve costs two instructions per step instead of
// a launch per panel (k_band_fwd serves the refinement solves).
__global__ void __launch_bounds__(64) k_band_panel(int32_t p, int32_t T, double* __restrict__ band, double* __restrict__ dfac,
                                                   double* __restrict__ sgn, double* __restrict__ invd,
                                                   int32_t* __restrict__ flag, double* __restrict__ wb,
                                                   double* __restrict__ wy) {
    const int lane = threadIdx.x, t = blockIdx.x;
    const int row = lane & (NB - 1);
    const bool below = lane >= NB;                                // a row below the diagonal tile
    double* Ap = band + (int64_t)p * (T + 1) * TS;
    double* mine = (below && t > 0) ? Ap + (int64_t)t * TS + row * NB : Ap + row * NB;
    double r[NB];
#pragma unroll
    for (int c = 0; c < NB; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(mine + c);
        r[c] = v.x;
        r[c + 1] = v.y;
    }
    double my_sign = 1.0, my_inv = 0.0;                           // lane k < 32: s_k, 1 / l_kk
    int32_t bad_at = 0, negative = 0;
    const int64_t brow = (int64_t)(p + ((below && t > 0) ? t : 0)) * NB + row;
    double bi = wb[brow], zi = 0.0;
    double dkk = lane_bcast(r[0], 0), piv, ipiv;
    pivot_roots(fabs(dkk) > 0.0 ? fabs(dkk) : 1.0, piv, ipiv);
    PanelRow st{my_sign, my_inv, bi, zi, bad_at, negative};
    panel_step<0>(r, st, lane, p, dkk, piv, ipiv);
    my_sign = st.sign;
    my_inv = st.inv;
    zi = st.zi;
    bi = st.bi;
    bad_at = st.bad_at;
    negative = st.negative;
    if (t == 0) {
        if (!below) {
            double* out = dfac + (int64_t)p * TS + row * NB;
#pragma unroll
            for (int c = 0; c < NB; c += 2)
                *reinterpret_cast<double2*>(out + c) = make_double2(c <= row ? r[c] : 0.0, c + 1 <= row ? r[c + 1] : 0.0);
            sgn[(int64_t)p * NB + row] = my_sign;
            invd[(int64_t)p * NB + row] = my_inv;
            wy[brow] = zi * my_sign;                              // S z_p: the right-hand side of the backward sweep
        }
        if (lane == 0) {                                          // panels run in stream order: the first bad pivot stays
            if (bad_at && flag[0] == 0) flag[0] = bad_at;
            if (negative) flag[1] += negative;
        }
        return;
    }
    if (below) {
#pragma unroll
        for (int c = 0; c < NB; c += 2) *reinterpret_cast<double2*>(mine + c) = make_double2(r[c], r[c + 1]);
        wb[brow] = bi;
    }
}

// trailing update of panel p: workgroup (i, j), 1 <= j <= i <= Tp: tile (p + j, i - j) -= L(p, i) S_p L(p, j)^T
// (2 x 2 tile pairs per workgroup -- a 64 x 64 x 32 product, 4 x 4 results per thread, two FMAs per LDS read -- were
// measured SLOWER at every size, 19.9 -> 24.8 ms at 915 panels of 18 tiles, 138 -> 146 ms at 2 793 panels of 91: the
// launch is over when its slowest workgroup is, and four times the work per workgroup costs more than the LDS reads
// saved; profiles/r04_direct_bench.txt)
__global__ void __launch_bounds__(256) k_band_update(int32_t p, int32_t T, double* __restrict__ band,
                                                     const double* __restrict__ sgn) {
    const int bi = blockIdx.x + 1, bj = blockIdx.y + 1;
    if (bj > bi) return;
    __shared__ double A[NB][NB + 1];
    __shared__ double B[NB][NB + 1];
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const double* Lp = band + (int64_t)p * (T + 1) * TS;
    // the four entries of the target tile this thread owns: requested before the operands are staged, so that their
    // latency is not paid after the product (each target tile belongs to exactly one workgroup of the launch)
    double* tgt = band + ((int64_t)(p + bj) * (T + 1) + (bi - bj)) * TS;
    const double t00 = tgt[ty * NB + tx], t01 = tgt[ty * NB + tx + 16], t10 = tgt[(ty + 16) * NB + tx],
                 t11 = tgt[(ty + 16) * NB + tx + 16];
    for (int e = tid; e < TS; e += 256) {
        A[e / NB][e % NB] = Lp[(int64_t)bi * TS + e] * sgn[(int64_t)p * NB + e % NB];   // the signs go into one operand
        B[e / NB][e % NB] = Lp[(int64_t)bj * TS + e];
    }
    __syncthreads();
    double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
#pragma unroll 8
    for (int k = 0; k < NB; ++k) {
        const double a0 = A[ty][k], a1 = A[ty + 16][k], b0 = B[tx][k], b1 = B[tx + 16][k];
        c00 += a0 * b0;
        c01 += a0 * b1;
        c10 += a1 * b0;
        c11 += a1 * b1;
    }
    tgt[ty * NB + tx] = t00 - c00;
    tgt[ty * NB + tx + 16] = t01 - c01;
    tgt[(ty + 16) * NB + tx] = t10 - c10;
    tgt[(ty + 16) * NB + tx + 16] = t11 - c11;
}

// ---- round 5: the same update on the f64 matrix cores.  v_mfma_f64_16x16x4_f64: D(16 x 16) += A(16 x 4) B(4 x 16), one f64
// of A and of B per lane (A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15]), four results per lane
// (D[row = (lane >> 4) + 4 reg][col = lane & 15]: cdna_hip_programming.md "f64 MFMA does NOT use these maps").  The f64
// matrix peak of MI355X equals its vector peak, so this is not about flops: the VALU product reads LDS once per FMA
// (4 ds_read_b64 per 4 v_fma_f64: the kernel is LDS-bound, 11 TFLOP/s at 91 tiles), an MFMA consumes 128 operand values
// per 2 reads per lane -- 8 x fewer LDS reads per flop -- and what is left is the traffic of the tiles themselves.
typedef double femcy_d4 __attribute__((ext_vector_type(4)));

// variant 1: one pair of panel tiles per workgroup as above, wave w = quadrant (w >> 1, w & 1) of the 32 x 32 target
// (bj_off: the launch covers target columns bj_off + 1 ... of the panel's update -- the two-stream schedule below splits
// the update into the column the next panel needs and the rest)
__global__ void __launch_bounds__(256) k_band_update_mfma(int32_t p, int32_t T, int32_t bj_off, double* __restrict__ band,
                                                          const double* __restrict__ sgn) {
    const int bi = blockIdx.x + 1, bj = blockIdx.y + 1 + bj_off;
    if (bj > bi) return;
    __shared__ double A[NB][NB + 1];
    __shared__ double B[NB][NB + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int qi = wave >> 1, qj = wave & 1, r16 = lane & 15, k4 = lane >> 4;
    const double* Lp = band + (int64_t)p * (T + 1) * TS;
    double* tgt = band + ((int64_t)(p + bj) * (T + 1) + (bi - bj)) * TS + (16 * qi + k4) * NB + 16 * qj + r16;
    double t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = tgt[4 * i * NB];                   // requested before the operands are staged
    for (int e = tid; e < TS; e += 256) {
        A[e / NB][e % NB] = Lp[(int64_t)bi * TS + e] * sgn[(int64_t)p * NB + e % NB];
        B[e / NB][e % NB] = Lp[(int64_t)bj * TS + e];
    }
    __syncthreads();
    femcy_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * qi + r16][4 * kk + k4], B[16 * qj + r16][4 * kk + k4], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) tgt[4 * i * NB] = t[i] - acc[i];
}

// variant 2: a 2 x 2 block of tile pairs per workgroup (a 64 x 64 x 32 product): four operand tiles staged once serve
// four products -- 20 KB of tile traffic per product instead of 32 -- wave w owns target tile (w >> 1, w & 1) of the block:
// 4 quadrants x 8 k-steps = 32 MFMAs.  (With VALU arithmetic this blocking lost in round 4: four times the LDS-bound
// work per workgroup; with the matrix cores the work per workgroup is ~0.2 us.)
__global__ void __launch_bounds__(256) k_band_update_mfma2(int32_t p, int32_t T, int32_t Tp, double* __restrict__ band,
                                                           const double* __restrict__ sgn) {
    const int I = blockIdx.x, J = blockIdx.y;
    if (J > I) return;
    __shared__ double A[2][NB][NB + 1];
    __shared__ double B[2][NB][NB + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1, r16 = lane & 15, k4 = lane >> 4;
    const int bi = 2 * I + 1 + wi, bj = 2 * J + 1 + wj;                       // this wave's tile pair
    const bool mine = bi <= Tp && bj <= bi;
    const double* Lp = band + (int64_t)p * (T + 1) * TS;
    double* tgt = band + ((int64_t)(p + bj) * (T + 1) + (bi - bj)) * TS + k4 * NB + r16;
    double t[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[a][b][i] = mine ? tgt[(16 * a + 4 * i) * NB + 16 * b] : 0.0;
    for (int e = tid; e < 2 * TS; e += 256) {
        const int w = e / TS, o = e % TS;
        const int ti = 2 * I + 1 + w, tj = 2 * J + 1 + w;
        A[w][o / NB][o % NB] = ti <= Tp ? Lp[(int64_t)ti * TS + o] * sgn[(int64_t)p * NB + o % NB] : 0.0;
        B[w][o / NB][o % NB] = tj <= Tp ? Lp[(int64_t)tj * TS + o] : 0.0;
    }
    __syncthreads();
    if (!mine) return;
    femcy_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = femcy_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double a0 = A[wi][r16][4 * kk + k4], a1 = A[wi][16 + r16][4 * kk + k4];
        const double b0 = B[wj][r16][4 * kk + k4], b1 = B[wj][16 + r16][4 * kk + k4];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) tgt[(16 * a + 4 * i) * NB + 16 * b] = t[a][b][i] - acc[a][b][i];
}

// forward substitution, panel p: z_p = L_pp^-1 b_p in the registers of the wave -- lane = row (both halves hold it), its
// row of L_pp in 32 registers, column-oriented: z_c goes from lane c to all, every later row subtracts its multiple.
// Workgroup 0 stores S z_p (the right-hand side of the backward sweep); workgroup g > 0 holds two tiles of the panel,
// tile 2 g - 1 in lanes 0..31 and tile 2 g in lanes 32..63 (lane = row, the row in registers), and subtracts
// L(p, t) z_p from the rows of block row p + t.
__global__ void __launch_bounds__(64) k_band_fwd(int32_t p, int32_t T, int32_t Tp, const double* __restrict__ band,
                                                 const double* __restrict__ dfac, const double* __restrict__ sgn,
                                                 const double* __restrict__ invd, double* __restrict__ wb,
                                                 double* __restrict__ wy) {
    const int lane = threadIdx.x, g = blockIdx.x;
    const int row = lane & (NB - 1), half = lane >> 5;
    const int tt = 2 * g - 1 + half;                              // my tile (g > 0)
    const bool tile = g > 0 && tt <= Tp;
    double d[NB], x[NB];
    const double* Dp = dfac + (int64_t)p * TS + row * NB;
    const double* Lt = band + ((int64_t)p * (T + 1) + (tile ? tt : 0)) * TS + row * NB;
#pragma unroll
    for (int c = 0; c < NB; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(Dp + c);
        d[c] = v.x;
        d[c + 1] = v.y;
        const double2 w = tile ? *reinterpret_cast<const double2*>(Lt + c) : make_double2(0.0, 0.0);
        x[c] = w.x;
        x[c + 1] = w.y;
    }
    double bi = wb[(int64_t)p * NB + row];
    const double iv = invd[(int64_t)p * NB + row];
    double zi = 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const double zc = lane_bcast(bi, c) * lane_bcast(iv, c);
        zi = (row == c) ? zc : zi;
        bi = (row > c) ? bi - d[c] * zc : bi;
    }
    if (g == 0) {
        if (half == 0) wy[(int64_t)p * NB + row] = zi * sgn[(int64_t)p * NB + row];
        return;
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) s += x[c] * lane_bcast(zi, c);
    if (tile) wb[(int64_t)(p + tt) * NB + row] -= s;
}

// backward substitution, panel p: x_p = L_pp^-T y_p, lane = column (its column of L_pp in registers); workgroup g > 0:
// tile t = 2 g - 1 / 2 g of panel p - t (the tiles whose rows are block row p), lane = column of the tile, subtracts
// L(p - t, t)^T x_p from y of panel p - t
__global__ void __launch_bounds__(64) k_band_bwd(int32_t p, int32_t T, int32_t Tq, const double* __restrict__ band,
                                                 const double* __restrict__ dfac, const double* __restrict__ invd,
                                                 double* __restrict__ wy, double* __restrict__ wx) {
    const int lane = threadIdx.x, g = blockIdx.x;
    const int col = lane & (NB - 1), half = lane >> 5;
    const int tt = 2 * g - 1 + half;
    const bool tile = g > 0 && tt <= Tq;
    double d[NB], x[NB];
    const double* Dp = dfac + (int64_t)p * TS + col;
    const double* Lt = band + ((int64_t)(p - (tile ? tt : 0)) * (T + 1) + (tile ? tt : 0)) * TS + col;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        d[r] = Dp[r * NB];                                        // L_pp(r, col)
        x[r] = tile ? Lt[r * NB] : 0.0;                           // L(p - t, t)(r, col)
    }
    double yi = wy[(int64_t)p * NB + col];
    const double iv = invd[(int64_t)p * NB + col];
    double xi = 0.0;
#pragma unroll
    for (int c = NB - 1; c >= 0; --c) {
        const double xc = lane_bcast(yi, c) * lane_bcast(iv, c);
        xi = (col == c) ? xc : xi;
        yi = (col < c) ? yi - d[c] * xc : yi;
    }
    if (g == 0) {
        if (half == 0) wx[(int64_t)p * NB + col] = xi;
        return;
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < NB; ++r) s += x[r] * lane_bcast(xi, r);
    if (tile) wy[(int64_t)(p - tt) * NB + col] -= s;
}

template <class Tp>
int alloc(Tp** p, size_t count) {
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    if (dmalloc(p, std::max<size_t>(count, 1) * sizeof(Tp)) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        set_error("direct solve: out of device memory (%zu bytes)", count * sizeof(Tp));
        return FEMCY_ENOMEM;
    }
    return FEMCY_OK;
}

// order + storage for the current pattern
// reverse Cuthill-McKee of the current pattern, computed once per pattern and kept on the host (femcy_direct_plan needs
// only this; femcy_direct_solve uploads it with the band)
const BandOrder& order_of(Ctx* c, DirectState& st) {
    if (st.order_serial != c->pattern_serial) {
        st.order = band_order_rcm(c->nn, c->ne, c->npe, c->h_elems.data());
        st.order_serial = c->pattern_serial;
    }
    return st.order;
}

int prepare(Ctx* c, DirectState& st) {
    if (st.pattern_serial == c->pattern_serial && st.d_band) return FEMCY_OK;
    const int64_t keep_limit = st.max_bytes;
    st.release();
    st.max_bytes = keep_limit;
    const BandOrder& o = order_of(c, st);
    st.half_band_nodes = o.half_band_nodes;
    const int64_t bw = ((int64_t)o.half_band_nodes + 1) * c->dm - 1;
    const int64_t T = (bw + NB - 1) / NB, P = (c->n + NB - 1) / NB;
    const double bytes = (double)P * (double)(T + 1) * TS * 8.0;
    if (bytes > (double)st.max_bytes) {
        set_error("direct solve: the band of this system (%lld DOF, %lld sub-diagonals after reverse Cuthill-McKee) takes "
                  "%.1f GB, more than the limit of %.1f GB (FEMCY_OPT_DIRECT_MAX_BYTES)",
                  (long long)c->n, (long long)bw, bytes * 1e-9, (double)st.max_bytes * 1e-9);
        return FEMCY_ENOMEM;
    }
    st.bw = (int32_t)bw;
    st.T = (int32_t)T;
    st.P = (int32_t)P;
    st.band_tiles = P * (T + 1);
    int rc;
    if ((rc = alloc(&st.d_rank, (size_t)c->nn)) || (rc = alloc(&st.d_node_at, (size_t)c->nn)) ||
        (rc = alloc(&st.d_band, (size_t)st.band_tiles * TS)) || (rc = alloc(&st.d_dfac, (size_t)P * TS)) ||
        (rc = alloc(&st.d_sgn, (size_t)P * NB)) || (rc = alloc(&st.d_invd, (size_t)P * NB)) || (rc = alloc(&st.d_wb, (size_t)P * NB)) ||
        (rc = alloc(&st.d_wy, (size_t)P * NB)) || (rc = alloc(&st.d_wx, (size_t)P * NB)) ||
        (rc = alloc(&st.d_res, (size_t)c->n)) || (rc = alloc(&st.d_Kx, (size_t)c->n)) || (rc = alloc(&st.d_norms, 2)) ||
        (rc = alloc(&st.d_flag, 2))) {
        st.release();
        st.max_bytes = keep_limit;
        return rc;
    }
    FEMCY_HIP(hipHostMalloc((void**)&st.h_back, 32, hipHostMallocDefault));
    FEMCY_HIP(hipMemcpy(st.d_rank, o.rank.data(), (size_t)c->nn * sizeof(int32_t), hipMemcpyHostToDevice));
    FEMCY_HIP(hipMemcpy(st.d_node_at, o.node_at.data(), (size_t)c->nn * sizeof(int32_t), hipMemcpyHostToDevice));
    st.pattern_serial = c->pattern_serial;
    return FEMCY_OK;
}

}  // namespace

void direct_release(Ctx* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_states.find(c);
    if (it != g_states.end()) {
        it->second.release();
        g_states.erase(it);
    }
}

int direct_set_update_variant(Ctx* c, int64_t v) {
    FEMCY_REQUIRE(v >= -1 && v <= 3, "direct solve, tile update: -1 (auto), 0 (VALU), 1 or 2 (matrix cores), 3 (1 on two streams)");
    state_of(c).update_variant = (int)v;
    return FEMCY_OK;
}

int direct_set_max_bytes(Ctx* c, int64_t bytes) {
    FEMCY_REQUIRE(bytes >= (int64_t)1 << 20, "direct solve: the band limit must be at least 1 MiB");
    state_of(c).max_bytes = bytes;
    return FEMCY_OK;
}

int direct_plan(Ctx* c, femcy_direct_info* info) {
    FEMCY_REQUIRE(info != nullptr, "femcy_direct_plan: info must not be null");
    FEMCY_REQUIRE(c->dm == 2 || c->dm == 3, "direct solve: dm = %d", c->dm);
    *info = femcy_direct_info{};
    DirectState& st = state_of(c);
    const BandOrder& o = order_of(c, st);
    const int64_t bw = ((int64_t)o.half_band_nodes + 1) * c->dm - 1;
    const int64_t T = (bw + NB - 1) / NB, P = (c->n + NB - 1) / NB;
    info->n = c->n;
    info->bandwidth = (int32_t)bw;
    info->panels = (int32_t)P;
    const double bytes = (double)P * (double)(T + 1) * TS * 8.0;
    info->band_bytes = bytes < 9.0e18 ? (int64_t)bytes : INT64_MAX;
    return FEMCY_OK;
}

int direct_solve(Ctx* c, const double* d_b, double* d_x, femcy_direct_info* info) {
    if (c->comm) {
        set_error("femcy_direct_solve: the factorisation is single-rank; a partitioned system is solved by femcy_pcg");
        return FEMCY_ECOMM;
    }
    FEMCY_REQUIRE(c->dm == 2 || c->dm == 3, "direct solve: dm = %d", c->dm);
    femcy_direct_info local{};
    if (!info) info = &local;
    *info = femcy_direct_info{};
    info->n = c->n;
    DirectState& st = state_of(c);
    int rc = prepare(c, st);
    if (rc) return rc;
    info->band_bytes = st.band_tiles * TS * 8;
    info->bandwidth = st.bw;
    info->panels = st.P;
    hipStream_t s = c->stream;
    const int32_t T = st.T, P = st.P;
    // ---- K -> band, factorisation
    FEMCY_HIP(hipMemsetAsync(st.d_band, 0, (size_t)st.band_tiles * TS * sizeof(double), s));
    FEMCY_HIP(hipMemsetAsync(st.d_flag, 0, 2 * sizeof(int32_t), s));
    {
        const int32_t maxL = *std::max_element(c->h_slice_len.begin(), c->h_slice_len.end());
        const int64_t threads = (int64_t)c->nslices * maxL * SLICE;
        const int grid = (int)((threads + 255) / 256);
        if (c->dm == 3)
            hipLaunchKernelGGL((k_band_fill<3>), dim3(grid), dim3(256), 0, s, c->nslices, maxL, c->d_slice_len, c->d_slice_off,
                               c->d_node_of, c->d_rowlen, c->d_bcol, c->d_Kvals, st.d_rank, T, st.d_band);
        else
            hipLaunchKernelGGL((k_band_fill<2>), dim3(grid), dim3(256), 0, s, c->nslices, maxL, c->d_slice_len, c->d_slice_off,
                               c->d_node_of, c->d_rowlen, c->d_bcol, c->d_Kvals, st.d_rank, T, st.d_band);
    }
    // band order of a right-hand side, the two sweeps, back to the caller's numbering (set or add)
    auto solve_into = [&](const double* d_rhs, double* d_out, int add, double* d_band_pad) -> int {
        hipLaunchKernelGGL(k_band_gather, dim3((P * NB + 255) / 256), dim3(256), 0, s, c->n, P, c->dm, T, st.d_node_at, d_rhs,
                           st.d_wb, d_band_pad);
        hipEvent_t prev_rest = nullptr;
        bool have_rest = false;
        const int want = st.update_variant >= 0 ? st.update_variant : (T >= DIRECT_OVERLAP_MIN_TILES ? DIRECT_OVERLAP_VARIANT : -1);
        if (want == 3 && !st.s2) {
            FEMCY_HIP(hipStreamCreateWithFlags(&st.s2, hipStreamNonBlocking));
            for (hipEvent_t& e : st.ev) FEMCY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (d_band_pad)   // first call: the factorisation sits between the padding of the band and the sweeps
            for (int32_t p = 0; p < P; ++p) {
                const int32_t Tp = std::min(T, P - 1 - p);
                hipLaunchKernelGGL(k_band_panel, dim3(Tp + 1), dim3(64), 0, s, p, T, st.d_band, st.d_dfac, st.d_sgn, st.d_invd,
                                   st.d_flag, st.d_wb, st.d_wy);
                if (Tp > 0) {
                    // 0 = VALU product (rounds 4-5 default below MFMA_MIN_TILES), 1 = matrix cores, one tile pair per
                    // workgroup, 2 = matrix cores, 2 x 2 tile pairs per workgroup
                    const int var = want >= 0 ? want : (Tp >= DIRECT_MFMA_MIN_TILES ? DIRECT_MFMA_VARIANT : 0);
                    if (var == 2)
                        hipLaunchKernelGGL(k_band_update_mfma2, dim3((Tp + 1) / 2, (Tp + 1) / 2), dim3(256), 0, s, p, T, Tp, st.d_band, st.d_sgn);
                    else if (var == 3 && Tp >= 2) {
                        // two-stream schedule (round 5): panel p + 1 needs only the tiles of column p + 1 updated, i.e. the
                        // bj = 1 column of this update; that part stays on the main stream, the rest (bj >= 2) runs on a
                        // second stream beside panel p + 1.  Orderings: rest(p) after panel(p) [its operands]; rest(p)
                        // after rest(p - 1) [same stream: they meet in the same target tiles]; first(p) after rest(p - 1)
                        // [both write column p + 1] -- rest(p) and first(p) write disjoint columns, rest(p) does not touch
                        // column p + 1, which panel p + 1 factors meanwhile.
                        hipEvent_t ep = st.ev[(2 * p) % DirectState::NEV], er = st.ev[(2 * p + 1) % DirectState::NEV];
                        FEMCY_HIP(hipEventRecord(ep, s));                                   // panel(p) done
                        if (have_rest) FEMCY_HIP(hipStreamWaitEvent(s, prev_rest, 0));      // rest(p - 1) done before first(p)
                        hipLaunchKernelGGL(k_band_update_mfma, dim3(Tp, 1), dim3(256), 0, s, p, T, 0, st.d_band, st.d_sgn);
                        FEMCY_HIP(hipStreamWaitEvent(st.s2, ep, 0));
                        hipLaunchKernelGGL(k_band_update_mfma, dim3(Tp, Tp - 1), dim3(256), 0, st.s2, p, T, 1, st.d_band, st.d_sgn);
                        FEMCY_HIP(hipEventRecord(er, st.s2));
                        prev_rest = er;
                        have_rest = true;
                    } else if (var == 1 || var == 3) {
                        if (have_rest) FEMCY_HIP(hipStreamWaitEvent(s, prev_rest, 0));      // (the last panels of a two-stream run)
                        hipLaunchKernelGGL(k_band_update_mfma, dim3(Tp, Tp), dim3(256), 0, s, p, T, 0, st.d_band, st.d_sgn);
                    }
                    else
                        hipLaunchKernelGGL(k_band_update, dim3(Tp, Tp), dim3(256), 0, s, p, T, st.d_band, st.d_sgn);
                }
            }
        if (have_rest) FEMCY_HIP(hipStreamWaitEvent(s, prev_rest, 0));   // the sweeps read what the second stream wrote last
        for (int32_t p = 0; p < P && !d_band_pad; ++p) {       // (the first solve's forward sweep rode with the panels)
            const int32_t Tp = std::min(T, P - 1 - p);
            hipLaunchKernelGGL(k_band_fwd, dim3(1 + (Tp + 1) / 2), dim3(64), 0, s, p, T, Tp, st.d_band, st.d_dfac, st.d_sgn,
                               st.d_invd, st.d_wb, st.d_wy);
        }
        for (int32_t p = P - 1; p >= 0; --p) {
            const int32_t Tq = std::min(T, p);
            hipLaunchKernelGGL(k_band_bwd, dim3(1 + (Tq + 1) / 2), dim3(64), 0, s, p, T, Tq, st.d_band, st.d_dfac, st.d_invd,
                               st.d_wy, st.d_wx);
        }
        hipLaunchKernelGGL(k_band_scatter, dim3((int)((c->n + 255) / 256)), dim3(256), 0, s, c->n, c->dm, st.d_node_at,
                           st.d_wx, d_out, add);
        return FEMCY_OK;
    };
    int32_t* h_flag = reinterpret_cast<int32_t*>(st.h_back);
    double* h_norms = reinterpret_cast<double*>(st.h_back + 16);
    // res = b - K x with the PCG's product; -> max|res| / max|b| (NaN stays NaN)
    auto residual = [&](double* rel) -> int {
        int rc2 = launch_spmv(c, d_x, st.d_Kx, nullptr, nullptr);
        if (rc2) return rc2;
        hipLaunchKernelGGL(k_band_residual, dim3(1), dim3(1024), 0, s, c->n, d_b, st.d_Kx, st.d_res, st.d_norms);
        FEMCY_HIP(hipMemcpyAsync(h_norms, st.d_norms, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        FEMCY_HIP(hipStreamSynchronize(s));
        *rel = (h_norms[0] != h_norms[0]) ? h_norms[0] : (h_norms[1] > 0.0 ? h_norms[0] / h_norms[1] : h_norms[0]);
        return FEMCY_OK;
    };
    // the pivot flags travel with the first residual: one host synchronisation per solve, not two (a singular matrix
    // costs a product nobody looks at)
    if ((rc = solve_into(d_b, d_x, 0, st.d_band))) return rc;
    FEMCY_HIP(hipMemcpyAsync(h_flag, st.d_flag, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    double rel = 0.0;
    if ((rc = residual(&rel))) return rc;
    FEMCY_HIP(hipGetLastError());
    info->negative_pivots = h_flag[1];
    if (h_flag[0]) {
        info->singular_at = h_flag[0];
        set_error("direct solve: pivot %d (band order) is zero or not a number -- the matrix is singular", h_flag[0] - 1);
        return FEMCY_ENUMERIC;
    }
    while (info->refinements < DIRECT_MAX_REFINE && rel > DIRECT_REFINE_ABOVE) {
        if ((rc = solve_into(st.d_res, d_x, 1, nullptr))) return rc;
        ++info->refinements;
        double rel2 = 0.0;
        if ((rc = residual(&rel2))) return rc;
        const bool stalled = !(rel2 < 0.5 * rel);
        rel = rel2;
        if (stalled) break;
    }
    FEMCY_HIP(hipGetLastError());
    info->residual = rel;
    if (!(rel <= DIRECT_ACCEPT)) {
        set_error("direct solve: residual %.3e max|b| after %d refinement steps (%d negative pivots): elimination without "
                  "pivoting lost this matrix", rel, info->refinements, info->negative_pivots);
        return FEMCY_ENUMERIC;
    }
    return FEMCY_OK;
}

}  // namespace femcy
