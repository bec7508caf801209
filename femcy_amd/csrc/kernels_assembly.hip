// Element-level kernels for gfx950: geometry (grad N, det J * w), deformation gradient + Cauchy stress,
// stiffness assembly into the blocked SELL-64 matrix, nodal-force gather, Dirichlet 0/1 elimination.
//
// Reference kernels replaced (paths relative to the FEMcy checkout):
//   get_dsdx_and_vol             stiffnessMtrx.py:132-150   -> k_geom
//   get_deformation_gradient     stiffnessMtrx.py:532-556   -> k_geom<STRESS=true>
//   constitutiveOfLargeDeform x4 material_zoo/*.py          -> cauchy_large()
//   assemble_stiffnessMtrx       stiffnessMtrx.py:161-186   -> k_assemble_gather / k_assemble_atomic
//   assemble_nodal_force_GN_kernel  stiffnessMtrx.py:620-644 -> k_nodal_force
//   dirichletBC_* kernels        stiffnessMtrx.py:279-341   -> k_dirichlet_zero
//
// Design notes (DESIGN.md has the long form):
//   * dN/dxi depends only on the Gauss point, so it is a per-type table read through the scalar
//     cache (wave-uniform index), never per element.
//   * K is never scattered with a search: the element->slot map is precomputed on the host.  The
//     default assembly is "owner computes": one lane per stored dm x dm block sums the contributions
//     of every (element, Gauss point) that touches it in a fixed order and writes the block once,
//     fully coalesced (8*nnz bytes reach HBM, no zero-fill pass, no atomics, bit-reproducible).
//     The atomic variant (f64 global_atomic_add) is kept for comparison and as the race check.
//   * B has 3 (3-D) / 2 (2-D) non-zeros per column; B^T C B is evaluated on those only, in the same
//     ascending Voigt order as the reference's dense products, so the only rounding differences are
//     FMA contraction and the (order-free in the reference) accumulation order over elements.
#include <algorithm>
#include <vector>
#include "ctx.hpp"
#include "element_math.hpp"

namespace femcy {

// (det_inv, det3, push_forward3, green_voigt3, cauchy_large: element_math.hpp -- shared with the host backend)

// a gradient row / force row / coordinate triple is dm contiguous doubles at 8-byte alignment: one 16-byte load (+ one
// 8-byte load for dm = 3) instead of dm 8-byte loads.  The gather kernels are bound by the texture-address path
// (24-byte gathers: DESIGN.md section 3); a third fewer load instructions is a third fewer address cycles.
typedef double femcy_d2u __attribute__((ext_vector_type(2), aligned(8)));
template <int DM>
__device__ __forceinline__ void load_row(const double* __restrict__ p, double (&v)[DM]) {
    const femcy_d2u t = *reinterpret_cast<const femcy_d2u*>(p);
    v[0] = t.x;
    v[1] = t.y;
    if (DM == 3) v[DM - 1] = p[2];
}

// records of W doubles held one per lane -> the workgroup's 256 consecutive records in global memory, written
// through an LDS transpose so that a wavefront stores 512 consecutive bytes per instruction instead of 64 8-byte
// pieces 8 W bytes apart (the element pass writes ~100..350 B per element; strided, the stores were its bottleneck)
template <int W>
__device__ __forceinline__ void block_store(double* __restrict__ dst, const double (&v)[W], int nvalid, double* lds) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W; ++i) lds[i * 257 + t] = v[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int q = k * 256 + t;
        if (q < nvalid * W) dst[q] = lds[(q % W) * 257 + q / W];
    }
    __syncthreads();
}

// the same for records that are not contiguous from one element to the next (several Gauss points per element) or wider
// than the staging area: W doubles of thread t go to dst[t * stride .. + W); a wavefront stores runs of W consecutive
// doubles (C3D10: 15 = 120 B) instead of 64 pieces `stride` doubles apart
template <int W>
__device__ __forceinline__ void block_store_strided(double* __restrict__ dst, const double* __restrict__ v, int nvalid,
                                                    double* lds, int64_t stride) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < W; ++i) lds[i * 257 + t] = v[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const int q = k * 256 + t;
        const int el = q / W, i = q - el * W;
        if (q < nvalid * W) dst[(int64_t)el * stride + i] = lds[i * 257 + el];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------ geometry
// one thread per element, Gauss points looped inside (dN[g] is then wave-uniform -> scalar loads)
template <int NPE, int DM, bool STRESS>
__global__ void __launch_bounds__(256) k_geom(int32_t ne, int32_t nGP, const double* __restrict__ nodes,
                                              const double* __restrict__ u, const int32_t* __restrict__ elems,
                                              const double* __restrict__ dN, const double* __restrict__ w,
                                              int mat_kind, const double* __restrict__ C, double p0, double p1,
                                              double* __restrict__ dsdx, double* __restrict__ vol,
                                              double* __restrict__ Fout, double* __restrict__ Sout,
                                              double* __restrict__ fe) {
    // staged stores (block_store) when an element's record is contiguous and small enough for the LDS transpose:
    // one Gauss point (C3D4, CPS3) for dsdx / F / sigma, always for the per-element nodal forces
    constexpr int WG = NPE * DM, WT = DM * DM;
    constexpr bool STAGE = WG <= 16;
    // round 3: wider records (C3D10: 30 doubles per Gauss point, four Gauss points) go through the same staging area in
    // chunks of 15 with an element stride -- 49 -> ~30 us of the C3D10 element pass were its 960-byte-strided stores.
    // round 6: so do the records of elements with several Gauss points whatever their width (CPE8: 16 doubles per Gauss
    // point, four Gauss points -- a lane's 128 contiguous bytes 512 bytes from its neighbour's: 64 partial lines per store
    // instruction), in chunks of the largest divisor of the record width up to 16
    constexpr int CHK = WG % 16 == 0 ? 16 : WG % 15 == 0 ? 15 : WG % 12 == 0 ? 12 : WG % 8 == 0 ? 8 : WG % 6 == 0 ? 6 : 1;
    constexpr bool WIDE = CHK > 1;
    __shared__ double stage_lds[STAGE ? 257 * (WG > CHK ? WG : CHK) : (WIDE ? 257 * CHK : 1)];
    const int32_t e0 = blockIdx.x * blockDim.x;
    const int nvalid = min(256, ne - e0);
    const bool valid = (int)threadIdx.x < nvalid;
    const int32_t e = valid ? e0 + (int32_t)threadIdx.x : ne - 1;      // idle lanes recompute the last element, store nothing
    const bool staged = STAGE && nGP == 1;
    double X[NPE][DM], U[NPE][DM];
    // per-element nodal forces fe[a][:] = sum_g gradN_a . sigma * vol (internal force only): the node gather then
    // reads dm doubles per incident element instead of a gradient row, a stress tensor and a weight
    double facc[STRESS ? NPE : 1][DM];
    if (STRESS) {
#pragma unroll
        for (int a = 0; a < NPE; ++a)
#pragma unroll
            for (int i = 0; i < DM; ++i) facc[a][i] = 0.0;
    }
#pragma unroll
    for (int a = 0; a < NPE; ++a) {
        const int32_t nd = elems[(int64_t)e * NPE + a];
        load_row<DM>(nodes + (int64_t)nd * DM, X[a]);
        if (u) {
            load_row<DM>(u + (int64_t)nd * DM, U[a]);
        } else {
#pragma unroll
            for (int i = 0; i < DM; ++i) U[a][i] = 0.0;
        }
    }
    for (int g = 0; g < nGP; ++g) {
        const double* __restrict__ dNg = dN + g * NPE * DM;
        double J[DM][DM], inv[DM][DM];
#pragma unroll
        for (int i = 0; i < DM; ++i)
#pragma unroll
            for (int j = 0; j < DM; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < NPE; ++a) acc += (X[a][i] + U[a][i]) * dNg[a * DM + j];
                J[i][j] = acc;
            }
        const double det = det_inv<DM>(J, inv);
        if (dsdx) {   // post-processing recomputes F / sigma without touching the stored geometry
            double G[WG];
#pragma unroll
            for (int a = 0; a < NPE; ++a)
#pragma unroll
                for (int j = 0; j < DM; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv[k][j];
                    G[a * DM + j] = acc;
                }
            if (staged) {
                if constexpr (STAGE) block_store<WG>(dsdx + (int64_t)e0 * WG, G, nvalid, stage_lds);
            } else if constexpr (WIDE) {
#pragma unroll
                for (int c = 0; c < WG / CHK; ++c)
                    block_store_strided<CHK>(dsdx + ((int64_t)e0 * nGP + g) * WG + c * CHK, G + c * CHK, nvalid, stage_lds,
                                             (int64_t)nGP * WG);
            } else if (valid) {
                double* out = dsdx + ((int64_t)e * nGP + g) * WG;
#pragma unroll
                for (int i = 0; i < WG; ++i) out[i] = G[i];
            }
            if (valid) vol[(int64_t)e * nGP + g] = det * w[g];
        }

        if (STRESS) {
            double J0[DM][DM], inv0[DM][DM], F[DM][DM], sig[DM][DM];
#pragma unroll
            for (int i = 0; i < DM; ++i)
#pragma unroll
                for (int j = 0; j < DM; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int a = 0; a < NPE; ++a) acc += X[a][i] * dNg[a * DM + j];
                    J0[i][j] = acc;
                }
            det_inv<DM>(J0, inv0);
#pragma unroll
            for (int i = 0; i < DM; ++i)
#pragma unroll
                for (int j = 0; j < DM; ++j) F[i][j] = 0.0;
#pragma unroll
            for (int a = 0; a < NPE; ++a) {
                double dsdX[DM];
#pragma unroll
                for (int j = 0; j < DM; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv0[k][j];
                    dsdX[j] = acc;
                }
#pragma unroll
                for (int i = 0; i < DM; ++i)
#pragma unroll
                    for (int j = 0; j < DM; ++j) F[i][j] += U[a][i] * dsdX[j];
            }
#pragma unroll
            for (int i = 0; i < DM; ++i) F[i][i] += 1.0;
            if (Fout) {   // the Newton residual needs neither F nor sigma in memory (femcy_residual_and_K)
                double Fl[WT];
#pragma unroll
                for (int i = 0; i < DM; ++i)
#pragma unroll
                    for (int j = 0; j < DM; ++j) Fl[i * DM + j] = F[i][j];
                if (staged) {
                    if constexpr (STAGE) block_store<WT>(Fout + (int64_t)e0 * WT, Fl, nvalid, stage_lds);
                } else if constexpr (WIDE) {
                    block_store_strided<WT>(Fout + ((int64_t)e0 * nGP + g) * WT, Fl, nvalid, stage_lds, (int64_t)nGP * WT);
                } else if (valid) {
                    double* fo = Fout + ((int64_t)e * nGP + g) * WT;
#pragma unroll
                    for (int i = 0; i < WT; ++i) fo[i] = Fl[i];
                }
            }
            if (Sout || fe) {   // get_deformation_gradient alone (post-processing) leaves the stored stress untouched
                cauchy_large<DM>(mat_kind, C, p0, p1, F, sig);
                if (Sout) {
                    double Sl[WT];
#pragma unroll
                    for (int i = 0; i < DM; ++i)
#pragma unroll
                        for (int j = 0; j < DM; ++j) Sl[i * DM + j] = sig[i][j];
                    if (staged) {
                        if constexpr (STAGE) block_store<WT>(Sout + (int64_t)e0 * WT, Sl, nvalid, stage_lds);
                    } else if constexpr (WIDE) {
                        block_store_strided<WT>(Sout + ((int64_t)e0 * nGP + g) * WT, Sl, nvalid, stage_lds, (int64_t)nGP * WT);
                    } else if (valid) {
                        double* so = Sout + ((int64_t)e * nGP + g) * WT;
#pragma unroll
                        for (int i = 0; i < WT; ++i) so[i] = Sl[i];
                    }
                }
                if (fe) {
                    const double vg = det * w[g];
#pragma unroll
                    for (int a = 0; a < NPE; ++a) {
                        double ga[DM];
#pragma unroll
                        for (int j = 0; j < DM; ++j) {
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv[k][j];
                            ga[j] = acc;
                        }
#pragma unroll
                        for (int i = 0; i < DM; ++i) {
                            double d = 0.0;
#pragma unroll
                            for (int j = 0; j < DM; ++j) d += ga[j] * sig[j][i];
                            facc[STRESS ? a : 0][i] += d * vg;
                        }
                    }
                }
            }
        }
    }
    if (STRESS && fe) {
        double Fe[WG];
#pragma unroll
        for (int a = 0; a < NPE; ++a)
#pragma unroll
            for (int i = 0; i < DM; ++i) Fe[a * DM + i] = facc[STRESS ? a : 0][i];
        if constexpr (STAGE) {
            block_store<WG>(fe + (int64_t)e0 * WG, Fe, nvalid, stage_lds);
        } else if constexpr (WIDE) {
#pragma unroll
            for (int c = 0; c < WG / CHK; ++c)
                block_store_strided<CHK>(fe + (int64_t)e0 * WG + c * CHK, Fe + c * CHK, nvalid, stage_lds, (int64_t)WG);
        } else if (valid) {
            double* out = fe + (int64_t)e * WG;
#pragma unroll
            for (int i = 0; i < WG; ++i) out[i] = Fe[i];
        }
    }
}

// (kblock_add: element_math.hpp)

// (kblock_consistent: element_math.hpp)

// owner-computes assembly of the consistent tangent: the per-block gather with F and sigma of each contributing
// Gauss point (an opt-in extension; the fast paths below assemble the reference's matrix)
template <int DM>
__global__ void __launch_bounds__(256) k_assemble_gather_consistent(int64_t npos, int32_t npe, int32_t nGP,
                                                                    const int32_t* __restrict__ ctr_ptr,
                                                                    const int32_t* __restrict__ ctr,
                                                                    const double* __restrict__ dsdx,
                                                                    const double* __restrict__ vol,
                                                                    const double* __restrict__ Fg,
                                                                    const double* __restrict__ Sg, bool neo,
                                                                    double lam, double mu, double p0, double p1,
                                                                    const int32_t* __restrict__ tpos, int skip_diag,
                                                                    double* __restrict__ Kvals) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    // the tangent is symmetric and its rows sum to zero as well (sum_b gradN_b = 0 in the material and in the
    // geometric part): upper blocks only, mirrored stores, diagonal from k_diag_from_rowsum -- see k_assemble_gather
    const int32_t tp = tpos[p];
    if (tp == -2 || (skip_diag && tp == (int32_t)p)) return;
    double acc[DM * DM];
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) acc[k] = 0.0;
    for (int32_t c = ctr_ptr[p]; c < ctr_ptr[p + 1]; ++c) {
        const int32_t code = ctr[c];
        const int32_t lb = code % npe;
        const int32_t t = code / npe;
        const int32_t la = t % npe;
        const int64_t e = t / npe;
        for (int g = 0; g < nGP; ++g) {
            const int64_t gp = e * nGP + g;
            const int64_t base = gp * npe;
            kblock_consistent<DM>(dsdx + (base + la) * DM, dsdx + (base + lb) * DM, Fg + gp * DM * DM, Sg + gp * DM * DM,
                                  neo, lam, mu, p0, p1, vol[gp], acc);
        }
    }
    const int64_t row = p >> 6;
    const int lane = (int)(p & 63);
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) Kvals[kv_index<DM>(row, k, lane)] = acc[k];
    if (tp >= 0 && tp != (int32_t)p) {
        const int64_t trow = tp >> 6;
        const int tlane = tp & 63;
#pragma unroll
        for (int r = 0; r < DM; ++r)
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) Kvals[kv_index<DM>(trow, cc * DM + r, tlane)] = acc[r * DM + cc];
    }
}

// (kblock_cubic3: element_math.hpp)

// owner-computes assembly: lane p = stored block (row = p / 64, lane = p % 64).
// (A variant with the element arity as a template parameter -- constant-divisor decode of the packed
// contribution code -- measured 28 % slower on gfx950 for C3D4 and equal for C3D10: the kernel is bound by
// L1 line throughput of the 24-byte dsdx gathers, not by the integer decode.)
// SYM: K is symmetric block-wise, K_ba = K_ab^T (C is symmetric).  Only the lanes of the diagonal and of the
// blocks with column node > row node evaluate their contributions; they also store the transpose at tpos[p],
// the position of block (b, a).  tpos = -2 marks the mirrored (skipped) lanes, -1 padding.  Halves the dsdx
// gathers that bound this kernel; the mirrored stores of a wavefront land in one block row of the neighbouring
// slice on structured numberings, i.e. they stay coalesced.  K comes out exactly symmetric.
template <int DM, bool SYM, bool CUBIC>
__global__ void __launch_bounds__(256) k_assemble_gather(int64_t npos, int32_t npe, int32_t nGP,
                                                         const int32_t* __restrict__ ctr_ptr,
                                                         const int32_t* __restrict__ ctr,
                                                         const int32_t* __restrict__ tpos,
                                                         const double* __restrict__ dsdx,
                                                         const double* __restrict__ vol, const double* __restrict__ C,
                                                         double c11, double c12, double c44,
                                                         double* __restrict__ Kvals, int skip_diag) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    int32_t tp = -1;
    if (SYM) {
        tp = tpos[p];
        if (tp == -2) return;
        if (skip_diag && tp == (int32_t)p) return;      // k_diag_from_rowsum fills the diagonal block afterwards
    }
    double acc[DM * DM];
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) acc[k] = 0.0;
    const int32_t c0 = ctr_ptr[p], c1 = ctr_ptr[p + 1];
    int32_t c = c0;
    // two contributions per trip: their (dependent) code -> dsdx -> FMA chains are independent, which doubles
    // the loads in flight per lane of this latency-bound gather; summation order stays ascending
    for (; c + 1 < c1; c += 2) {
        const int32_t code0 = ctr[c], code1 = ctr[c + 1];
        const int32_t lb0 = code0 % npe, t0 = code0 / npe, la0 = t0 % npe;
        const int32_t lb1 = code1 % npe, t1 = code1 / npe, la1 = t1 % npe;
        const int64_t e0 = t0 / npe, e1 = t1 / npe;
        double acc1[DM * DM];
#pragma unroll
        for (int k = 0; k < DM * DM; ++k) acc1[k] = 0.0;
        for (int g = 0; g < nGP; ++g) {
            const int64_t b0 = (e0 * nGP + g) * npe, b1 = (e1 * nGP + g) * npe;
            double ga0[DM], gb0[DM], ga1[DM], gb1[DM];
            load_row<DM>(dsdx + (b0 + la0) * DM, ga0);
            load_row<DM>(dsdx + (b0 + lb0) * DM, gb0);
            load_row<DM>(dsdx + (b1 + la1) * DM, ga1);
            load_row<DM>(dsdx + (b1 + lb1) * DM, gb1);
            if constexpr (CUBIC && DM == 3) {
                kblock_cubic3(ga0, gb0, c11, c12, c44, vol[e0 * nGP + g], acc);
                kblock_cubic3(ga1, gb1, c11, c12, c44, vol[e1 * nGP + g], acc1);
            } else {
                kblock_add<DM>(ga0, gb0, C, vol[e0 * nGP + g], acc);
                kblock_add<DM>(ga1, gb1, C, vol[e1 * nGP + g], acc1);
            }
        }
#pragma unroll
        for (int k = 0; k < DM * DM; ++k) acc[k] += acc1[k];
    }
    for (; c < c1; ++c) {
        const int32_t code = ctr[c];
        const int32_t lb = code % npe;
        const int32_t t = code / npe;
        const int32_t la = t % npe;
        const int64_t e = t / npe;
        for (int g = 0; g < nGP; ++g) {
            const int64_t base = (e * nGP + g) * npe;
            double ga[DM], gb[DM];
            load_row<DM>(dsdx + (base + la) * DM, ga);
            load_row<DM>(dsdx + (base + lb) * DM, gb);
            if constexpr (CUBIC && DM == 3)
                kblock_cubic3(ga, gb, c11, c12, c44, vol[e * nGP + g], acc);
            else
                kblock_add<DM>(ga, gb, C, vol[e * nGP + g], acc);
        }
    }
    const int64_t row = p >> 6;
    const int lane = (int)(p & 63);
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) Kvals[kv_index<DM>(row, k, lane)] = acc[k];
    if (SYM && tp >= 0 && tp != (int32_t)p) {
        const int64_t trow = tp >> 6;
        const int tlane = tp & 63;
#pragma unroll
        for (int r = 0; r < DM; ++r)
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) Kvals[kv_index<DM>(trow, cc * DM + r, tlane)] = acc[r * DM + cc];
    }
}

// diagonal blocks from the row-sum identity.  sum_b gradN_b = 0 (partition of unity; checked on the element tables in
// femcy_set_element) makes sum_b B_b = 0, hence sum_b K_ab = 0 for every row of the assembled (pre-Dirichlet) matrix:
// K_aa = - sum_{b != a} K_ab.  The diagonal block collects all ~24 incident elements of a C3D4 node -- 40 % of the
// gathers of the symmetric assembly -- and is obtained here from one coalesced pass over the row instead.
template <int DM>
__global__ void __launch_bounds__(256) k_diag_from_rowsum(int32_t nslices, const int32_t* __restrict__ node_of,
                                                          const int32_t* __restrict__ rowlen,
                                                          const int64_t* __restrict__ slice_off,
                                                          double* __restrict__ Kvals) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // position = slice * 64 + lane
    if (p >= (int64_t)nslices * SLICE) return;
    const int32_t a = node_of[p];
    if (a < 0) return;
    const int64_t off = slice_off[p >> 6];
    const int lane = (int)(p & 63);
    const int32_t L = rowlen[a];
    double acc[DM * DM];
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) acc[k] = 0.0;
    for (int32_t j = 1; j < L; ++j)
#pragma unroll
        for (int k = 0; k < DM * DM; ++k) acc[k] -= Kvals[kv_index<DM>(off + j, k, lane)];
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) Kvals[kv_index<DM>(off, k, lane)] = acc[k];
}

// row-centric assembly: one wavefront per node (matrix block row).  Lanes are the (incident element, local
// column node) pairs of the row -- 24 x 4 = 96 for an interior C3D4 node -- so the four lanes of one element read
// one contiguous 96-byte dsdx record (1 cache line per element instead of 1 per lane), every lane evaluates
// exactly one B_a^T C B_b block (no trip-count imbalance between diagonal and off-diagonal blocks), and the row
// is reduced in a wave-private LDS accumulator with ds_add_f64 before it is written out.  Deterministic: the
// accumulator is private to the wave and LDS serves conflicting lanes of one instruction in lane order.
template <int DM>
__global__ void __launch_bounds__(256) k_assemble_rows(int32_t nn, int32_t npe, int32_t nGP, int32_t Lmax,
                                                       const int32_t* __restrict__ ne_ptr,
                                                       const int32_t* __restrict__ ne_idx,
                                                       const uint16_t* __restrict__ slotj,
                                                       const int32_t* __restrict__ rowlen,
                                                       const int32_t* __restrict__ pos,
                                                       const int64_t* __restrict__ slice_off,
                                                       const double* __restrict__ dsdx,
                                                       const double* __restrict__ vol, const double* __restrict__ C,
                                                       double* __restrict__ Kvals) {
    extern __shared__ __attribute__((aligned(16))) double lds_rows[];
    constexpr int DD = DM * DM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* acc = lds_rows + (size_t)wave * Lmax * DD;
    for (int32_t base = blockIdx.x * 4; base < nn; base += gridDim.x * 4) {   // uniform trip count per workgroup
        const int32_t a = base + wave;
        const bool valid = a < nn;
        const int32_t L = valid ? rowlen[a] : 0;
        for (int idx = lane; idx < L * DD; idx += 64) acc[idx] = 0.0;
        __syncthreads();
        if (valid) {
            const int32_t k0 = ne_ptr[a];
            const int32_t ntask = (ne_ptr[a + 1] - k0) * npe;
            for (int32_t t = lane; t < ntask; t += 64) {
                const int32_t kk = t / npe, lb = t - kk * npe;
                const int32_t code = ne_idx[k0 + kk];          // e*npe + la
                const int64_t e = code / npe;
                const int32_t la = code - (int32_t)e * npe;
                const int32_t j = slotj[(int64_t)code * npe + lb];
                double blk[DD];
#pragma unroll
                for (int k = 0; k < DD; ++k) blk[k] = 0.0;
                for (int g = 0; g < nGP; ++g) {
                    const int64_t row = (e * nGP + g) * npe;
                    kblock_add<DM>(dsdx + (row + la) * DM, dsdx + (row + lb) * DM, C, vol[e * nGP + g], blk);
                }
#pragma unroll
                for (int k = 0; k < DD; ++k) atomicAdd(&acc[j * DD + k], blk[k]);
            }
        }
        __syncthreads();
        if (valid) {
            const int32_t pa = pos[a];
            const int64_t off = slice_off[pa >> 6];
            const int lanea = pa & 63;
            for (int idx = lane; idx < L * DD; idx += 64) {
                const int j = idx / DD, k = idx - j * DD;
                Kvals[kv_index<DM>(off + j, k, lanea)] = acc[idx];
            }
        }
        __syncthreads();
    }
}

// LDS hand-off between the lanes of ONE wavefront: the LDS serves a wave's instructions in issue order, so all that
// is needed is that the compiler keeps the order (no s_barrier: the waves of a workgroup run independently here)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// row-centric assembly, second form (3-D elements whose shape-function gradients sum to zero; default for C3D10).
// One workgroup per 64-row slice of the SELL matrix, one wavefront per row at a time (16 rows per wavefront).
// Differences to k_assemble_rows, each answering a counter of profiles/r02_pmc_c3d10_baseline.txt:
//   * the dsdx / vol records of the row's incident elements are staged in wave-private LDS by full-width 16-byte
//     loads (one 1-KiB instruction per C3D10 element) instead of 28 scattered 8-byte gathers per lane and Gauss
//     point -- the old kernel kept the texture-address path 60 % busy (TA_TA_BUSY) with 22 cache lines per load
//     instruction and spent 65 % of its wave cycles in s_waitcnt;
//   * a pass = up to EPC incident elements of one row; the records of pass p+1 and the element list of pass p+2
//     are fetched into registers while pass p computes (a row's chain row -> element list -> records is three
//     dependent memory round trips; un-pipelined they, not the arithmetic, set the kernel time);
//   * lanes are the (element, column node != row node) pairs, 7 elements x 9 columns = 63 lanes per pass for
//     C3D10; the diagonal block, which every incident element hits (7-way ds_add_f64 conflicts), is not
//     accumulated at all but taken from the row sum K_aa = -sum_{b != a} K_ab at the end, in LDS;
//   * cubic-pattern C (all reference materials): 30 instead of 90 multiply-adds per block and Gauss point;
//   * all rows of a slice are written by one workgroup (one XCD's L2).  (Writing the four rows the waves of a
//     workgroup finish together as 64-byte requests, behind two barriers per row, was measured too: TA busy 140 M ->
//     82 M cycles, WRITE_SIZE 517 -> 439 MB, kernel 389 -> 401 us -- the barriers cost what the requests saved.
//     A fourth wave per SIMD -- __launch_bounds__(256, 4), LDS accumulator sized per slice-length class -- measured
//     605 us: the 128-VGPR cap spills the prefetch registers.  XCD-contiguous slice ranges as in k_spmv (for record
//     reuse inside one L2: FETCH_SIZE is 563 MB reported = 1.1 GB for 123 MB of records) measured 500 us: FETCH only
//     fell to 485 MB and the corner-node slices, three passes per row, all land on the first XCDs.  Rows in
//     element-major order (nodes sorted by first incident element, XCD-contiguous ranges of that order): 590 us, FETCH
//     still 486 MB, WRITE 832 MB -- so the fetch is not record re-reads but the read-for-fill of the K lines that
//     are written 16 bytes at a time; what bounds the kernel besides LDS and VALU is ~0.9 GB of fabric traffic for
//     a 357 MB matrix, and only writing whole 128-byte lines (eight adjacent rows at once) would remove it.)
// Deterministic: fixed pass order, ds_add_f64 of one instruction applied in lane order, fixed-order diagonal sum.
// Where its time goes (profiles/r03_rows2_probe.txt; build with -DFEMCY_ROWS2_PROBE=<bits>, results meaningless): bit 1
// plain LDS stores instead of atomics, 2 no LDS reads / arithmetic, 4 no LDS block writes, 8 no record loads, 16 no
// global stores, 32 no row end, 64 no staging writes.
#ifdef FEMCY_ROWS2_PROBE
#define ROWS2_PROBE_BIT(b_) ((FEMCY_ROWS2_PROBE & (b_)) != 0)
#else
#define ROWS2_PROBE_BIT(b_) 0
#endif
template <int NPE, int NGP, bool CUBIC>
__global__ void __launch_bounds__(256) k_assemble_rows2(int32_t nslices, int32_t Lmax,
                                                        const int32_t* __restrict__ ne_ptr,
                                                        const int32_t* __restrict__ ne_idx,
                                                        const uint16_t* __restrict__ slotj,
                                                        const int32_t* __restrict__ rowlen,
                                                        const int32_t* __restrict__ node_of,
                                                        const int64_t* __restrict__ slice_off,
                                                        const double* __restrict__ dsdx, const double* __restrict__ vol,
                                                        const double* __restrict__ C, double c11, double c12, double c44,
                                                        double* __restrict__ Kvals) {
    constexpr int DM = 3, DD = 9, T = NPE - 1, EPC = 64 / T, RD = NGP * NPE * DM, P16 = RD / 2;
    constexpr int NIT = (EPC * P16 + 63) / 64;                   // 16-byte pieces per lane and pass (C3D10: 7)
    constexpr int VOLW = (EPC * NGP + 1) & ~1, CODEW = (EPC + 1) / 2 * 2 / 2 + 1;
    static_assert(RD % 2 == 0, "records are staged in 16-byte pieces");
    static_assert(EPC * NGP <= 64, "one vol value per lane and pass");
    extern __shared__ __attribute__((aligned(16))) double lds_rows2[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int accw = (Lmax * DD + 1) & ~1;
    double* rec = lds_rows2 + (size_t)wave * (EPC * RD + VOLW + accw + 2 * CODEW);
    double* vl = rec + EPC * RD;
    double* acc = vl + VOLW;
    int32_t* codes = reinterpret_cast<int32_t*>(acc + accw);
    const int32_t s = blockIdx.x;
    if (s >= nslices) return;
    const int64_t off = slice_off[s];

    // ---- the wave's rows (slice lanes wave, wave + 4, ...): lane i holds the metadata of its i-th row
    constexpr int RPW = SLICE / 4;
    int32_t m_L = 0, m_k0 = 0, m_cnt = 0;
    bool m_valid = false;
    if (lane < RPW) {
        const int32_t a = node_of[(int64_t)s * SLICE + wave + 4 * lane];
        if (a >= 0) {                                            // padding lanes only at the tail of the last slice
            m_valid = true;
            m_L = rowlen[a];
            m_k0 = ne_ptr[a];
            m_cnt = ne_ptr[a + 1] - m_k0;
        }
    }
    const int nrows = __popcll(__ballot(m_valid));               // valid rows are a prefix
    if (nrows == 0) return;
#define ROW_CNT(i) __builtin_amdgcn_readlane(m_cnt, (i))
#define ROW_K0(i) __builtin_amdgcn_readlane(m_k0, (i))
#define ROW_L(i) __builtin_amdgcn_readlane(m_L, (i))
    // pass iterator: (row i, first incident element c0); a row without elements still takes one (empty) pass
    auto advance = [&](int& i, int& c0) {
        c0 += EPC;
        if (i < nrows && c0 >= ROW_CNT(i)) {
            ++i;
            c0 = 0;
        }
    };
    auto load_codes = [&](int i, int c0) -> int32_t {           // lane q: (element, local row node) code of element q
        if (i >= nrows) return 0;
        const int32_t nE = min(EPC, ROW_CNT(i) - c0);
        return lane < nE ? ne_idx[ROW_K0(i) + c0 + lane] : 0;
    };
    double2 R[NIT];
    double V = 0.0;
    int32_t JS = 0;          // slot of the lane's block in the row (needed last, fetched with the records: VMEM returns
                             // in order, so a load issued in the compute phase would wait for the prefetches before it)
    // 16-byte pieces of a pass's records + det J w into registers (a macro, not a lambda: R must stay in VGPRs)
#define LOAD_RECORDS(code_, i_, c0_)                                                                  \
    if ((i_) < nrows) {                                                                               \
        const int32_t nE_ = min(EPC, ROW_CNT(i_) - (c0_));                                            \
        _Pragma("unroll") for (int u = 0; u < NIT; ++u) {                                             \
            const int32_t p_ = lane + 64 * u;                                                         \
            const int32_t q_ = p_ / P16, w_ = p_ - q_ * P16;                                          \
            const int64_t e_ = __shfl((code_), q_, 64) / NPE;                                         \
            if (p_ < nE_ * P16 && !ROWS2_PROBE_BIT(8)) R[u] = reinterpret_cast<const double2*>(dsdx + e_ * RD)[w_];   \
        }                                                                                             \
        const int32_t qv_ = lane / NGP, gv_ = lane - qv_ * NGP;                                       \
        const int64_t ev_ = __shfl((code_), qv_, 64) / NPE;                                           \
        if (lane < nE_ * NGP) V = vol[ev_ * NGP + gv_];                                               \
        const int32_t qt_ = lane / T, jb_ = lane - qt_ * T;                                           \
        const int32_t ct_ = __shfl((code_), qt_, 64);                                                 \
        const int32_t lat_ = ct_ % NPE;                                                               \
        if (lane < nE_ * T) JS = slotj[(int64_t)ct_ * NPE + jb_ + (jb_ >= lat_ ? 1 : 0)];             \
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) R[u] = make_double2(0.0, 0.0);

    int i0 = 0, c00 = 0, i1 = 0, c01 = 0, i2 = 0, c02 = 0;
    advance(i1, c01);
    i2 = i1;
    c02 = c01;
    advance(i2, c02);
    int32_t code_c = load_codes(i0, c00);
    int32_t code_n = load_codes(i1, c01);
    LOAD_RECORDS(code_c, i0, c00)
    for (int idx = lane; idx < ROW_L(0) * DD; idx += 64) acc[idx] = 0.0;

    while (i0 < nrows) {
        const int32_t cnt = ROW_CNT(i0);
        const int32_t nE = max(0, min(EPC, cnt - c00));
        wave_lds_sync();                                        // the previous pass is done with rec / vl / codes
        if (lane < nE) codes[lane] = code_c;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int32_t p = lane + 64 * u;
            const int32_t q = p / P16, w = p - q * P16;
            if (p < nE * P16 && !ROWS2_PROBE_BIT(64)) reinterpret_cast<double2*>(rec + q * RD)[w] = R[u];
        }
        if (lane < nE * NGP) vl[lane] = V;
        int32_t j = JS;
        asm volatile("" : "+v"(j));     // consume JS HERE: a copy the compiler sinks below the prefetch would have to wait
                                        // for the prefetch loads as well (vmcnt counts in order)
        // prefetch: records of the next pass (its element list arrived during the previous pass), list of the one after
        LOAD_RECORDS(code_n, i1, c01)
        const int32_t code_nn = load_codes(i2, c02);
        wave_lds_sync();
        if (lane < nE * T) {
            const int32_t q = lane / T, jb = lane - q * T;
            const int32_t code = codes[q];
            const int32_t la = code % NPE;
            const int32_t lb = jb + (jb >= la ? 1 : 0);         // every column node of the element but the row node
            double blk[DD];
#pragma unroll
            for (int k = 0; k < DD; ++k) blk[k] = 0.0;
#pragma unroll
            for (int g = 0; g < NGP; ++g) {
                const double* ga = rec + q * RD + (g * NPE + la) * DM;
                const double* gb = rec + q * RD + (g * NPE + lb) * DM;
#if ROWS2_PROBE_BIT(2)
                blk[g] += (double)(q + lb);                           // probe: no LDS reads, no arithmetic
#else
                if (CUBIC) outer3_add(ga, gb, vl[q * NGP + g], blk);   // the geometric sum; constants at the end of the row
                else kblock_add<3>(ga, gb, C, vl[q * NGP + g], blk);
#endif
            }
#if ROWS2_PROBE_BIT(1)
#pragma unroll
            for (int k = 0; k < DD; ++k) acc[j * DD + k] = blk[k];   // probe: plain LDS stores instead of atomics
#elif ROWS2_PROBE_BIT(4)
            if (blk[0] == 1.2345) acc[j] = blk[1] + blk[2] + blk[3] + blk[4] + blk[5] + blk[6] + blk[7] + blk[8];   // probe: no LDS writes
#else
#pragma unroll
            for (int k = 0; k < DD; ++k) atomicAdd(&acc[j * DD + k], blk[k]);
#endif
        }
        if (c00 + EPC >= cnt && !ROWS2_PROBE_BIT(32)) {         // last pass of the row: diagonal, constants, write-out
            const int32_t L = ROW_L(i0);
            const int r = wave + 4 * i0;
            wave_lds_sync();
            // diagonal block from the row sum: 7 x 9 partial sums over the slots 1.., combined in a fixed order
            // (rec is free: the pass above only read it before the sync)
            if (lane < 63) {
                const int jj = lane / DD, k = lane - jj * DD;
                double t = 0.0;
                for (int32_t j = 1 + jj; j < L; j += 7) t += acc[j * DD + k];
                rec[lane] = t;
            }
            wave_lds_sync();
            if (lane < DD) {
                double t = 0.0;
#pragma unroll
                for (int jj = 0; jj < 7; ++jj) t += rec[jj * DD + lane];
                acc[lane] = -t;
            }
            wave_lds_sync();
            if (CUBIC) {                                        // S -> K, one stored block per lane, in place
                for (int32_t jb = lane; jb < L; jb += 64) {
                    double S[DD], Kb[DD];
#pragma unroll
                    for (int k = 0; k < DD; ++k) S[k] = acc[jb * DD + k];
                    cubic_from_outer3(S, c11, c12, c44, Kb);
#pragma unroll
                    for (int k = 0; k < DD; ++k) acc[jb * DD + k] = Kb[k];
                }
                wave_lds_sync();
            }
            // the row: per block four 16-byte pairs + the trailing 8-byte entry (kv_index layout), lane r of the slice
            double* __restrict__ Krow = Kvals + off * (int64_t)(DD * SLICE);
            for (int idx = lane; idx < L * 5; idx += 64) {
                const int j = idx / 5, pc = idx - j * 5;
                double* dst = Krow + (int64_t)j * (DD * SLICE);
#if ROWS2_PROBE_BIT(16)
                if (pc < 4) acc[j * DD + 2 * pc] += (double)(r + (dst - Krow));     // probe: no store instructions
#else
                if (pc < 4) {
                    reinterpret_cast<double2*>(dst + pc * (2 * SLICE))[r] =
                        make_double2(acc[j * DD + 2 * pc], acc[j * DD + 2 * pc + 1]);
                } else {
                    dst[4 * (2 * SLICE) + r] = acc[j * DD + 8];
                }
#endif
            }
            wave_lds_sync();
            if (i0 + 1 < nrows)
                for (int idx = lane; idx < ROW_L(i0 + 1) * DD; idx += 64) acc[idx] = 0.0;
        }
        i0 = i1; c00 = c01;
        i1 = i2; c01 = c02;
        advance(i2, c02);
        code_c = code_n;
        code_n = code_nn;
    }
#undef LOAD_RECORDS
#undef ROW_CNT
#undef ROW_K0
#undef ROW_L
}

// row-centric assembly, third form (round 3): k_assemble_rows2 with the write-out of EIGHT adjacent rows at a time.
// rows2 writes a finished row as 16-byte pieces, one per block entry pair, 1 KiB apart (the lane-interleaved SELL
// layout is made for the product): every 128-byte line of K is touched by eight rows at eight different times, the
// L2 fetches the line for the first partial write (read-for-fill) and often writes it back more than once --
// profiles/r02_pmc_rows2_c3d10.txt: FETCH 563 MB + WRITE 515 MB for 357 MB of K + 123 MB of records.  Here wave w
// of the workgroup takes rows 8 g + 2 w and 8 g + 2 w + 1 of group g (two accumulators per wave), the four waves meet
// at the end of the group, and all 256 threads write the group: eight lanes x 16 bytes = one whole 128-byte line per
// (block, entry pair), 64 bytes for the trailing entry plane -- no line is written twice, none is fetched.  Two
// workgroup barriers per 8 rows (rows2's experiment with 4-row / 64-byte write-outs paid two per row).
// Everything else (LDS-staged records, the three-deep software pipeline over passes, lanes = (element, column != row)
// pairs, row-sum diagonal, 30-flop blocks for cubic C, fixed order => bit-reproducible) is rows2's.
template <int NPE, int NGP, bool CUBIC>
__global__ void __launch_bounds__(256) k_assemble_rows3(int32_t nslices, int32_t nn, int32_t Lmax,
                                                        const int32_t* __restrict__ ne_ptr,
                                                        const int32_t* __restrict__ ne_idx,
                                                        const uint16_t* __restrict__ slotj,
                                                        const int32_t* __restrict__ rowlen,
                                                        const int32_t* __restrict__ node_of,
                                                        const int64_t* __restrict__ slice_off,
                                                        const double* __restrict__ dsdx, const double* __restrict__ vol,
                                                        const double* __restrict__ C, double c11, double c12, double c44,
                                                        double* __restrict__ Kvals) {
    constexpr int DM = 3, DD = 9, T = NPE - 1, EPC = 64 / T, RD = NGP * NPE * DM, P16 = RD / 2;
    constexpr int NIT = (EPC * P16 + 63) / 64;                   // 16-byte pieces per lane and pass (C3D10: 7)
    constexpr int VOLW = (EPC * NGP + 1) & ~1, CODEW = (EPC + 1) / 2 * 2 / 2 + 1;
    static_assert(RD % 2 == 0, "records are staged in 16-byte pieces");
    static_assert(EPC * NGP <= 64, "one vol value per lane and pass");
    extern __shared__ __attribute__((aligned(16))) double lds_rows3[];
    __shared__ int32_t gL[8];                                    // row lengths of the group being written
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int accw = (Lmax * DD + 1) & ~1;
    const int wstride = EPC * RD + VOLW + 2 * accw + 2 * CODEW;  // doubles per wave
    double* rec = lds_rows3 + (size_t)wave * wstride;
    double* vl = rec + EPC * RD;
    double* acc0 = vl + VOLW;                                    // two row accumulators: rows 2 w and 2 w + 1 of a group
    int32_t* codes = reinterpret_cast<int32_t*>(acc0 + 2 * accw);
    const int32_t s = blockIdx.x;
    if (s >= nslices) return;
    const int64_t off = slice_off[s];
    const int32_t nvalid = min(SLICE, nn - s * SLICE);           // valid rows are a prefix of the slice
    const int ngroups = (nvalid + 7) / 8;

    // ---- the wave's rows: its i-th row is slice lane 8 (i / 2) + 2 wave + (i % 2); lane i holds the metadata
    constexpr int RPW = SLICE / 4;
    int32_t m_L = 0, m_k0 = 0, m_cnt = 0;
    bool m_valid = false;
    if (lane < RPW) {
        const int32_t a = node_of[(int64_t)s * SLICE + 8 * (lane >> 1) + 2 * wave + (lane & 1)];
        if (a >= 0) {
            m_valid = true;
            m_L = rowlen[a];
            m_k0 = ne_ptr[a];
            m_cnt = ne_ptr[a + 1] - m_k0;
        }
    }
    const int nrows = __popcll(__ballot(m_valid));               // valid rows are a prefix in i as well
#define ROW_CNT(i) __builtin_amdgcn_readlane(m_cnt, (i))
#define ROW_K0(i) __builtin_amdgcn_readlane(m_k0, (i))
#define ROW_L(i) __builtin_amdgcn_readlane(m_L, (i))
    auto advance = [&](int& i, int& c0) {
        c0 += EPC;
        if (i < nrows && c0 >= ROW_CNT(i)) {
            ++i;
            c0 = 0;
        }
    };
    auto load_codes = [&](int i, int c0) -> int32_t {           // lane q: (element, local row node) code of element q
        if (i >= nrows) return 0;
        const int32_t nE = min(EPC, ROW_CNT(i) - c0);
        return lane < nE ? ne_idx[ROW_K0(i) + c0 + lane] : 0;
    };
    double2 R[NIT];
    double V = 0.0;
    int32_t JS = 0;
#define LOAD_RECORDS(code_, i_, c0_)                                                                  \
    if ((i_) < nrows) {                                                                               \
        const int32_t nE_ = min(EPC, ROW_CNT(i_) - (c0_));                                            \
        _Pragma("unroll") for (int u = 0; u < NIT; ++u) {                                             \
            const int32_t p_ = lane + 64 * u;                                                         \
            const int32_t q_ = p_ / P16, w_ = p_ - q_ * P16;                                          \
            const int64_t e_ = __shfl((code_), q_, 64) / NPE;                                         \
            if (p_ < nE_ * P16) R[u] = reinterpret_cast<const double2*>(dsdx + e_ * RD)[w_];          \
        }                                                                                             \
        const int32_t qv_ = lane / NGP, gv_ = lane - qv_ * NGP;                                       \
        const int64_t ev_ = __shfl((code_), qv_, 64) / NPE;                                           \
        if (lane < nE_ * NGP) V = vol[ev_ * NGP + gv_];                                               \
        const int32_t qt_ = lane / T, jb_ = lane - qt_ * T;                                           \
        const int32_t ct_ = __shfl((code_), qt_, 64);                                                 \
        const int32_t lat_ = ct_ % NPE;                                                               \
        if (lane < nE_ * T) JS = slotj[(int64_t)ct_ * NPE + jb_ + (jb_ >= lat_ ? 1 : 0)];             \
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) R[u] = make_double2(0.0, 0.0);

    int i0 = 0, c00 = 0, i1 = 0, c01 = 0, i2 = 0, c02 = 0;
    advance(i1, c01);
    i2 = i1;
    c02 = c01;
    advance(i2, c02);
    int32_t code_c = load_codes(i0, c00);
    int32_t code_n = load_codes(i1, c01);
    LOAD_RECORDS(code_c, i0, c00)

    for (int g = 0; g < ngroups; ++g) {
        // both accumulators of the wave start at zero (the previous group has been written: second barrier below)
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * g + h;
            if (i < nrows) {
                double* acc = acc0 + h * accw;
                for (int idx = lane; idx < ROW_L(i) * DD; idx += 64) acc[idx] = 0.0;
            }
        }
        while (i0 < nrows && (i0 >> 1) == g) {
            double* acc = acc0 + (i0 & 1) * accw;
            const int32_t cnt = ROW_CNT(i0);
            const int32_t nE = max(0, min(EPC, cnt - c00));
            wave_lds_sync();                                        // the previous pass is done with rec / vl / codes
            if (lane < nE) codes[lane] = code_c;
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int32_t p = lane + 64 * u;
                const int32_t q = p / P16, w = p - q * P16;
                if (p < nE * P16) reinterpret_cast<double2*>(rec + q * RD)[w] = R[u];
            }
            if (lane < nE * NGP) vl[lane] = V;
            int32_t j = JS;
            asm volatile("" : "+v"(j));     // consume JS here (see k_assemble_rows2)
            LOAD_RECORDS(code_n, i1, c01)
            const int32_t code_nn = load_codes(i2, c02);
            wave_lds_sync();
            if (lane < nE * T) {
                const int32_t q = lane / T, jb = lane - q * T;
                const int32_t code = codes[q];
                const int32_t la = code % NPE;
                const int32_t lb = jb + (jb >= la ? 1 : 0);         // every column node of the element but the row node
                double blk[DD];
#pragma unroll
                for (int k = 0; k < DD; ++k) blk[k] = 0.0;
#pragma unroll
                for (int gp = 0; gp < NGP; ++gp) {
                    const double* ga = rec + q * RD + (gp * NPE + la) * DM;
                    const double* gb = rec + q * RD + (gp * NPE + lb) * DM;
                    if (CUBIC) kblock_cubic3(ga, gb, c11, c12, c44, vl[q * NGP + gp], blk);
                    else kblock_add<3>(ga, gb, C, vl[q * NGP + gp], blk);
                }
#pragma unroll
                for (int k = 0; k < DD; ++k) atomicAdd(&acc[j * DD + k], blk[k]);
            }
            if (c00 + EPC >= cnt) {                                 // last pass of the row: diagonal from the row sum
                const int32_t L = ROW_L(i0);
                wave_lds_sync();
                if (lane < 63) {
                    const int jj = lane / DD, k = lane - jj * DD;
                    double t = 0.0;
                    for (int32_t jx = 1 + jj; jx < L; jx += 7) t += acc[jx * DD + k];
                    rec[lane] = t;
                }
                wave_lds_sync();
                if (lane < DD) {
                    double t = 0.0;
#pragma unroll
                    for (int jj = 0; jj < 7; ++jj) t += rec[jj * DD + lane];
                    acc[lane] = -t;
                }
                if (lane == 0) gL[2 * wave + (i0 & 1)] = L;
            }
            i0 = i1; c00 = c01;
            i1 = i2; c01 = c02;
            advance(i2, c02);
            code_c = code_n;
            code_n = code_nn;
        }
        // ---- the group's eight rows are complete: all threads write them, whole lines at a time
        __syncthreads();
        {
            const int r8 = threadIdx.x & 7;                          // row of the group
            const int rr = 8 * g + r8;                               // slice lane
            const bool rv = rr < nvalid;
            const int32_t Lr = rv ? gL[r8] : 0;
            int32_t Lg = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) Lg = max(Lg, (8 * g + q < nvalid) ? gL[q] : 0);
            const double* accr = lds_rows3 + (size_t)(r8 >> 1) * wstride + (EPC * RD + VOLW) + (r8 & 1) * accw;
            double* __restrict__ Krow = Kvals + off * (int64_t)(DD * SLICE);
            for (int idx = threadIdx.x >> 3; idx < Lg * 5; idx += 32) {
                const int j = idx / 5, pc = idx - j * 5;
                double* dst = Krow + (int64_t)j * (DD * SLICE);
                const bool has = j < Lr;                             // rows shorter than the group's longest: zero blocks
                if (rv) {
                    if (pc < 4) {
                        reinterpret_cast<double2*>(dst + pc * (2 * SLICE))[rr] =
                            has ? make_double2(accr[j * DD + 2 * pc], accr[j * DD + 2 * pc + 1]) : make_double2(0.0, 0.0);
                    } else {
                        dst[4 * (2 * SLICE) + rr] = has ? accr[j * DD + 8] : 0.0;
                    }
                }
            }
        }
        __syncthreads();
    }
#undef LOAD_RECORDS
#undef ROW_CNT
#undef ROW_K0
#undef ROW_L
}

// scatter assembly with hardware f64 atomics: one lane per element-local (a,b) block
template <int DM>
__global__ void __launch_bounds__(256) k_assemble_atomic(int64_t npair, int32_t npe, int32_t nGP,
                                                         const int32_t* __restrict__ elems,
                                                         const uint16_t* __restrict__ slotj,
                                                         const int32_t* __restrict__ pos,
                                                         const int64_t* __restrict__ slice_off,
                                                         const double* __restrict__ dsdx,
                                                         const double* __restrict__ vol, const double* __restrict__ C,
                                                         double* __restrict__ Kvals) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npair) return;
    const int32_t lb = (int32_t)(t % npe);
    const int64_t q = t / npe;
    const int32_t la = (int32_t)(q % npe);
    const int64_t e = q / npe;
    double acc[DM * DM];
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) acc[k] = 0.0;
    for (int g = 0; g < nGP; ++g) {
        const int64_t base = (e * nGP + g) * npe;
        kblock_add<DM>(dsdx + (base + la) * DM, dsdx + (base + lb) * DM, C, vol[e * nGP + g], acc);
    }
    const int32_t pa = pos[elems[e * npe + la]];
    const int64_t row = slice_off[pa >> 6] + slotj[t];
    const int lane = pa & 63;
#pragma unroll
    for (int k = 0; k < DM * DM; ++k) unsafeAtomicAdd(&Kvals[kv_index<DM>(row, k, lane)], acc[k]);
}

// row-centric assembly, fourth form (round 3): TWO rows per wavefront at a time, half a wave (32 lanes) each.
// profiles/r03_rows2_probe.txt: rows2 spends ~1 000 instructions per matrix row -- a wave runs the whole pass
// machinery (element lists, record staging, row end) for ONE row whose typical pass fills 45 of its 64 lanes and most
// of whose control is scalar -- and it is bound by the number of instructions the CU can issue, not by arithmetic,
// bandwidth or latency.  Here every instruction of the machinery serves two rows:
//   * lanes 0..31 own row i, lanes 32..63 row i + 1 of the wave's share of the slice (rows sorted by length: the two
//     have the same number of incident elements or nearly); a step takes up to three incident elements per row: lane
//     (q, t) = (gl / 10, gl % 10) computes the block of column node t of element q -- the diagonal block included
//     (t = the row node), so there is no row-sum pass and the element tables need not sum to zero;
//   * within a row the ten lanes of an element hit ten different slots, the three elements rarely the same one:
//     ds_add_f64 without the 7-way conflicts the diagonal had in a 64-lane pass;
//   * the records of a step (6 x 16 bytes per lane) are prefetched during the step before, as in rows2;
//   * at the end of a pair of rows each lane takes one stored block (32 per trip: one trip for most rows) out of LDS,
//     leaves zeros, applies the material constants and stores its five pieces.
// Deterministic for the same reason as rows2 (fixed step order, ds_add_f64 of one instruction applied in lane order).
//
// GP > 0 (round 5, the verdict's "a wave owns adjacent rows and writes whole lines from its own LDS tile"; experiment,
// FEMCY_TUNE_ROWS4_TILE = 1000 GP + LCUT): in slices no wider than `lcut` blocks a wave owns 16 CONSECUTIVE rows, keeps the finished
// rows of GP pairs (2 GP adjacent rows) in a tile of its own LDS and writes them out together: a store instruction
// then covers 32 GP contiguous bytes of 32 / GP slots instead of 32 bytes of 32 slots -- no workgroup barrier (rows3
// lost 57 us to one).  Wider slices (the corner nodes' rows) run as before.
template <int NPE, int NGP, bool CUBIC, int GP>
__global__ void __launch_bounds__(256) k_assemble_rows4(int32_t nslices, int32_t Lmax, int32_t lcut, int32_t wstride,
                                                        int32_t xcdc, const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ ne_ptr,
                                                        const int32_t* __restrict__ ne_idx,
                                                        const uint16_t* __restrict__ slotj,
                                                        const int32_t* __restrict__ rowlen,
                                                        const int32_t* __restrict__ node_of,
                                                        const int64_t* __restrict__ slice_off,
                                                        const double* __restrict__ dsdx, const double* __restrict__ vol,
                                                        const double* __restrict__ C, double c11, double c12, double c44,
                                                        double* __restrict__ Kvals) {
    constexpr int DM = 3, DD = 9, G = 32, EPG = G / NPE, RD = NGP * NPE * DM, P16 = RD / 2;
    constexpr int NIT = P16 / NPE;                               // 16-byte pieces per lane and step (C3D10: 6): lane (q, t)
                                                                 // fetches and stages pieces t, t + 10, ... of ITS element
                                                                 // q -- one address per lane and step, the rest immediates
    constexpr int VOLW = (EPG * NGP + 1) & ~1;
    static_assert(RD % 2 == 0 && EPG == 3 && P16 % NPE == 0 && NGP <= NPE, "half-wave layout: three elements per row and step");
    extern __shared__ __attribute__((aligned(16))) double lds_rows4[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 5, gl = lane & 31, gbase = grp * G;
    const int q = gl / NPE, t = gl - q * NPE;                    // element of the step, column node
    const int accw = (Lmax * DD + 1) & ~1;
    constexpr int RG = GP ? 2 * GP : 2;                          // rows of a tile
    const int accw_t = (lcut * DD) | 1;                          // an odd number of doubles: the rows of a tile start in different banks
    double* wbase = lds_rows4 + (size_t)wave * wstride;
    // workgroup b takes the b-th slice in order of decreasing work (pattern.cpp: `asm_order`).  Workgroups are dispatched
    // in index order and run on XCD b % 8 (observed; speed only): longest first gives every XCD an even share of every
    // weight class and lets the short slices fill the tail.  Round 4: in plain slice order a row order whose long rows
    // recur with a period that is a multiple of 8 slices (the coordinate orders of FEMCY_OPT_NODE_ORDER) put twice the
    // work on one XCD -- 303 -> 483 us with identical instruction counts, profiles/r04_pmc_rows4_node_order.txt --
    // and contiguous per-XCD ranges (balanced by blocks: 392 us, by work: 355 us) lose the mixing of long and short
    // slices the caller's numbering happens to have (its corner rows come first)
    // round 6 (`xcdc`, with `order` = the slices in Morton order of their centroids): workgroup b takes entry b / 8 of XCD
    // (b % 8)'s CONTIGUOUS eighth of the order -- on a mesh whose element records exceed the Infinity Cache (C3D10 k = 12:
    // 0.99 GB) the kernel was bound by re-fetching every record once per node of its element from HBM (FETCH 5.9 GB
    // reported for 2.8 GB of K, profiles/r06_pmc_c3d10_k12_first.txt); neighbouring slices on one L2 find them there
    int32_t wgi = (int32_t)blockIdx.x;
    if (xcdc) wgi = (wgi & 7) * ((int32_t)gridDim.x >> 3) + (wgi >> 3);
    if (wgi >= nslices) return;
    const int32_t s = __builtin_amdgcn_readfirstlane(order[wgi]);
    const int64_t off_v = slice_off[s];
    const int64_t off = ((int64_t)__builtin_amdgcn_readfirstlane((int32_t)(off_v >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)off_v);
    const bool tile = GP > 0 && __builtin_amdgcn_readfirstlane((int32_t)(slice_off[s + 1] - off_v)) <= lcut;   // the slice's width
    double* rec = wbase + (size_t)grp * (tile ? EPG * RD + VOLW + 2 : EPG * RD + VOLW + accw + 2);     // this half's records
    double* vl = rec + EPG * RD;
    double* acc = vl + VOLW;                                     // pair layout: this half's row
    double* tbase = wbase + 2 * (EPG * RD + VOLW + 2);           // tile layout: RG rows of accw_t

    // ---- the wave's rows: pair b = slice lanes 8 b + 2 wave and + 1 -- ADJACENT rows, so that the two halves of a
    // store instruction write neighbouring 16-byte pieces of the same line (one request instead of two), and the eight
    // rows the four waves finish at about the same time fill whole 128-byte lines; lane i holds the metadata of the
    // wave's i-th row (pair i / 2, half i % 2)
    constexpr int RPW = SLICE / 4;
    int32_t m_L = 0, m_k0 = 0, m_cnt = 0;
    bool m_valid = false;
    if (lane < RPW) {
        const int32_t a = node_of[(int64_t)s * SLICE + (tile ? RPW * wave + lane : 8 * (lane >> 1) + 2 * wave + (lane & 1))];
        if (a >= 0) {
            m_valid = true;
            m_L = rowlen[a];
            m_k0 = ne_ptr[a];
            m_cnt = ne_ptr[a + 1] - m_k0;
        }
    }
    // valid rows are a prefix of the slice: of this wave's rows too (its row index grows with i)
    const int nrows = __popcll(__ballot(m_valid));
    if (nrows == 0) return;
    const int npairs = (nrows + 1) / 2;
#define R4_CNT(i) __builtin_amdgcn_readlane(m_cnt, (i))
    // steps of pair b: enough for the longer of its two element lists, at least one (a row without elements is still
    // written)
    auto steps_of = [&](int b) -> int {
        const int32_t c = max(R4_CNT(2 * b), R4_CNT(2 * b + 1));           // rows >= nrows: cnt = 0
        return max(1, (c + EPG - 1) / EPG);
    };
    auto advance = [&](int& b, int& c) {
        ++c;
        if (b < npairs && c >= steps_of(b)) {
            ++b;
            c = 0;
        }
    };
    // per-lane view of step (b, c): elements of MY row in this step
    // a value of row 2 b (lower half) / 2 b + 1 (upper half) of the metadata lanes: two scalar reads and a select.
    // NO cross-lane traffic through the LDS in the loop: a step of the first version made twelve ds_bpermute round
    // trips one after the other (each waits for the one before: ~1 400 cycles per step, 118 of the kernel's 334 us with
    // everything else compiled out, profiles/r03_rows4_probe.txt)
    // (both scalar reads first, then the select: a readlane inside an arm of ?: is compiled as a divergent branch)
    auto rowval = [&](int32_t m, int b) -> int32_t {             // b <= npairs + 4 <= 12: lane index < 64
        const int32_t lo = __builtin_amdgcn_readlane(m, 2 * b), hi = __builtin_amdgcn_readlane(m, 2 * b + 1);
        return grp ? hi : lo;
    };
#define R4_ROWVAL(m_, b_) rowval((m_), (b_))
    auto my_nE = [&](int b, int c) -> int32_t {                  // pairs beyond the last: rows with cnt = 0
        const int32_t cnt = R4_ROWVAL(m_cnt, b);
        return max(0, min(EPG, cnt - EPG * c));
    };
    // Every global load of the loop is UNCONDITIONAL (lanes without work read entry 0 and the value is dropped) and
    // no loaded value is touched before the step that needs it: only then are the compiler's waits counted
    // (s_waitcnt vmcnt(N), N = the loads of the step in between) instead of vmcnt(0), and a step waits for the loads
    // issued TWO steps earlier while the batch issued one step earlier stays in flight.  With ~300 instructions per
    // step the loop is latency-bound (profiles/r03_rows4_probe.txt): the second batch in flight is what pays here
    // (it did not in rows2, which is bound by its instruction count).
    auto load_codes = [&](int b, int c) -> int32_t {             // lane gl < nE of each half: (element, local row node) code
        const int32_t nE = my_nE(b, c);
        const int32_t k0 = R4_ROWVAL(m_k0, b);
        return ne_idx[gl < nE ? k0 + EPG * c + gl : 0];          // other lanes: any valid code, never used
    };
    double2 RA[NIT], RB[NIT];
    double VA = 0.0, VB = 0.0;
    int32_t JA = 0, JB = 0;
    // the code of MY element (q) of a step: lane gbase + q of the list register -- ONE ds_bpermute per use (twelve
    // dependent ones per step were the first version's problem, six scalar reads + selects cost 20 VALU slots)
#define R4_MYCODE(code_) __shfl((code_), gbase + q, 64)
#define R4_LOAD_RECORDS(R_, V_, J_, code_, b_, c_)                                                    \
    {                                                                                                 \
        const int32_t nE_ = my_nE((b_), (c_));                                                        \
        const bool ok_ = q < nE_;                                                                     \
        const int32_t kq_ = ok_ ? R4_MYCODE(code_) : 0;                  /* my element's code; 0 = any valid one */ \
        const int64_t eq_ = kq_ / NPE;                                                                \
        const double2* rb_ = reinterpret_cast<const double2*>(dsdx + eq_ * RD) + t;                   \
        _Pragma("unroll") for (int u = 0; u < NIT; ++u)                                               \
            if (!ROWS2_PROBE_BIT(8)) R_[u] = rb_[NPE * u];                                            \
        V_ = vol[eq_ * NGP + (t < NGP ? t : 0)];                                                      \
        J_ = slotj[(int64_t)kq_ * NPE + t];                                                           \
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) RA[u] = RB[u] = make_double2(0.0, 0.0);

    // steps p .. p + 4: (pair of rows, step inside the pair)
    int b0 = 0, c00 = 0, b1 = 0, c01 = 0, b2, c02, b3, c03, b4, c04;
    advance(b1, c01);
    b2 = b1; c02 = c01; advance(b2, c02);
    b3 = b2; c03 = c02; advance(b3, c03);
    b4 = b3; c04 = c03; advance(b4, c04);
    int32_t code0 = load_codes(b0, c00), code1 = load_codes(b1, c01), cnewA = load_codes(b2, c02),
            cnewB = load_codes(b3, c03);
    R4_LOAD_RECORDS(RA, VA, JA, code0, b0, c00)
    R4_LOAD_RECORDS(RB, VB, JB, code1, b1, c01)
    if (tile) {
        for (int idx = lane; idx < RG * accw_t; idx += 64) tbase[idx] = 0.0;
    } else {
        for (int idx = gl; idx < accw; idx += G) acc[idx] = 0.0; // every row leaves the slots it used zeroed
    }

    auto compute = [&](int32_t nE, int32_t j, int32_t la, double* __restrict__ acc) {
        if (gl < nE * NPE) {
            double blk[DD];
#pragma unroll
            for (int k = 0; k < DD; ++k) blk[k] = 0.0;
#pragma unroll
            for (int g = 0; g < NGP; ++g) {
                const double* ga = rec + q * RD + (g * NPE + la) * DM;
                const double* gb = rec + q * RD + (g * NPE + t) * DM;
#if ROWS2_PROBE_BIT(2)
                blk[g] += (double)(q + t + la);                        // probe: no LDS reads, no arithmetic
#else
                if (CUBIC) outer3_add(ga, gb, vl[q * NGP + g], blk);   // geometric sum; constants at the end of the row
                else kblock_add<3>(ga, gb, C, vl[q * NGP + g], blk);
#endif
            }
#if ROWS2_PROBE_BIT(1)
#pragma unroll
            for (int k = 0; k < DD; ++k) acc[j * DD + k] = blk[k];   // probe: plain LDS stores instead of atomics
#elif ROWS2_PROBE_BIT(4)
            if (blk[0] == 1.2345) acc[j] = blk[1] + blk[2] + blk[3] + blk[4] + blk[5] + blk[6] + blk[7] + blk[8];   // probe: no LDS writes
#else
#pragma unroll
            for (int k = 0; k < DD; ++k) atomicAdd(&acc[j * DD + k], blk[k]);
#endif
        }
    };
    auto pair_end = [&](int b) {                                // the pair is complete: constants, write-out, zeros
        double* __restrict__ Krow = Kvals + off * (int64_t)(DD * SLICE);
        if (GP > 0 && tile) {
            constexpr int GPD = GP ? GP : 1;
            if ((b % GPD) != GPD - 1 && b != npairs - 1) return; // the tile is not complete yet
            const int g0 = (b / GPD) * GPD;                      // its first pair
            int32_t Lg = 0;
#pragma unroll
            for (int i = 0; i < RG; ++i) Lg = max(Lg, __builtin_amdgcn_readlane(m_L, 2 * g0 + i));    // rows >= nrows: 0
            const int rt = lane % RG, r = RPW * wave + 2 * g0 + rt;
            double* trow = tbase + rt * accw_t;
            wave_lds_sync();                                     // the atomics of the last step have landed
            for (int32_t jb = lane / RG; jb < Lg; jb += 64 / RG) {      // shorter rows of the tile: zero blocks, as stored
                double S[DD], Kb[DD];
#pragma unroll
                for (int k = 0; k < DD; ++k) S[k] = trow[jb * DD + k];
#pragma unroll
                for (int k = 0; k < DD; ++k) trow[jb * DD + k] = 0.0;
                if (CUBIC) cubic_from_outer3(S, c11, c12, c44, Kb);
                else {
#pragma unroll
                    for (int k = 0; k < DD; ++k) Kb[k] = S[k];
                }
                double* dst = Krow + (int64_t)jb * (DD * SLICE);
#pragma unroll
                for (int pc = 0; pc < 4; ++pc)
                    reinterpret_cast<double2*>(dst + pc * (2 * SLICE))[r] = make_double2(Kb[2 * pc], Kb[2 * pc + 1]);
                dst[4 * (2 * SLICE) + r] = Kb[8];
            }
            return;
        }
        const int32_t L = R4_ROWVAL(m_L, b);                    // rows >= nrows: 0
        const int r = 8 * b + 2 * wave + grp;
        int32_t ln = gl;
        asm volatile("" : "+v"(ln));    // addresses built on the lane are computed HERE, not hoisted into VGPRs for the
                                        // whole kernel
        wave_lds_sync();                                        // the atomics of the last step have landed
        for (int32_t jb = ln; jb < L; jb += G) {
            double S[DD], Kb[DD];
#pragma unroll
            for (int k = 0; k < DD; ++k) S[k] = acc[jb * DD + k];
#pragma unroll
            for (int k = 0; k < DD; ++k) acc[jb * DD + k] = 0.0;
            if (CUBIC) cubic_from_outer3(S, c11, c12, c44, Kb);
            else {
#pragma unroll
                for (int k = 0; k < DD; ++k) Kb[k] = S[k];
            }
#if ROWS2_PROBE_BIT(16)
            if (Kb[0] + Kb[1] + Kb[2] + Kb[3] + Kb[4] + Kb[5] + Kb[6] + Kb[7] + Kb[8] == 1.2345) acc[jb * DD] = (double)r;   // probe: no stores
#else
            double* dst = Krow + (int64_t)jb * (DD * SLICE);
#ifdef FEMCY_ROWS4_NT_STORES    /* experiment: K written past the L2 (profiles/r03_rows4_probe.txt) */
            typedef double d2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                d2v v2 = {Kb[2 * pc], Kb[2 * pc + 1]};
                __builtin_nontemporal_store(v2, reinterpret_cast<d2v*>(dst + pc * (2 * SLICE)) + r);
            }
            __builtin_nontemporal_store(Kb[8], dst + 4 * (2 * SLICE) + r);
#else
#pragma unroll
            for (int pc = 0; pc < 4; ++pc)
                reinterpret_cast<double2*>(dst + pc * (2 * SLICE))[r] = make_double2(Kb[2 * pc], Kb[2 * pc + 1]);
            dst[4 * (2 * SLICE) + r] = Kb[8];
#endif
#endif
        }
    };
    // one step: stage set X (records of step p) into LDS, refill X with the records of step p + 2 (their element list
    // arrived two steps ago), fetch the list of step p + 4, compute step p
#define R4_STEP(R_, V_, J_, CN_)                                                                      \
    {                                                                                                 \
        int32_t code2 = CN_;                                    /* list of step p + 2, fetched in step p - 2 */ \
        asm volatile("" : "+v"(code2));                         /* its own register: the refill of CN_ is not copied at the latch */ \
        const int32_t nE = my_nE(b0, c00);                                                            \
        const int32_t la = R4_MYCODE(code0) % NPE;              /* the row node inside my element of this step */ \
        wave_lds_sync();                                        /* the previous step is done with rec / vl */ \
        if (q < nE && !ROWS2_PROBE_BIT(64)) {                                                         \
            double2* sb_ = reinterpret_cast<double2*>(rec + q * RD) + t;                              \
            _Pragma("unroll") for (int u = 0; u < NIT; ++u) sb_[NPE * u] = R_[u];                     \
            if (t < NGP) vl[q * NGP + t] = V_;                                                        \
        }                                                                                             \
        int32_t j = J_;                                                                               \
        asm volatile("" : "+v"(j));     /* take the copy HERE, before the refill overwrites the register */ \
        CN_ = load_codes(b4, c04);                                                                    \
        R4_LOAD_RECORDS(R_, V_, J_, code2, b2, c02)                                                   \
        wave_lds_sync();                                                                              \
        compute(nE, j, la, (GP > 0 && tile) ? tbase + ((b0 % (GP ? GP : 1)) * 2 + grp) * accw_t : acc); \
        if (b0 < npairs && c00 + 1 >= steps_of(b0) && !ROWS2_PROBE_BIT(32)) pair_end(b0);             \
        b0 = b1; c00 = c01; b1 = b2; c01 = c02; b2 = b3; c02 = c03; b3 = b4; c03 = c04;               \
        advance(b4, c04);                                                                             \
        code0 = code1; code1 = code2;                                                                 \
    }
    // ONE exit, at the latch (a break between the halves becomes an edge from the first half to the loop header in
    // the structured control flow, and the path-insensitive wait-count analysis then drains everything there); a step
    // beyond the last pair -- the second half, at most once per wave -- is empty
    while (b0 < npairs) {
        R4_STEP(RA, VA, JA, cnewA)
        R4_STEP(RB, VB, JB, cnewB)
    }
#undef R4_STEP
#undef R4_MYCODE
#undef R4_ROWVAL
#undef R4_LOAD_RECORDS
#undef R4_CNT
}

// row-centric assembly, fifth form (round 6; FEMCY_ASM_PAIRS): written for the 2-D quadratic families, whose rows are
// SHORT (a CPE8 corner node has 4 incident elements and 21 blocks, a mid-side node 2 and 13) and MANY (1 M-DOF beam:
// 494 k rows) -- the opposite of C3D10.  The generic k_assemble_rows spends a wavefront, two barriers and 28 scattered
// 8-byte gathers per lane on each of them (profiles/r06_pmc_asm_cpe8_rows.txt); rows2 / rows4 pay ~1 000 instructions of
// pass machinery per row.  Here:
//   * a wavefront owns a CHUNK of 16 adjacent rows of a slice (workgroup = slice) and walks the chunk's (row, incident
//     element) pairs, listed in storage order by the host (ensure_pairs): one scalar load for the chunk's range, one
//     coalesced load for up to 64 pair codes, then the records -- three dependent round trips per chunk, not per row;
//   * NPE lanes per pair, lane (q, t) = column node t of pair q: its NGP gradient loads ARE the element's record, each
//     byte fetched once per pair, 128 contiguous bytes per Gauss point and pair (CPE8); the row node's gradients are
//     lane (q, la)'s and det J w lane (q, g)'s: ds_bpermute, no second fetch, no staging pass;
//   * a block is linear in the geometric sum S_ab = sum_g |J| w (grad N_a (x) grad N_b) whatever the (mesh-wide) C is:
//     K_ab[i][k] = sum_jl C[v(i,j)][v(k,l)] S[j][l].  Lanes accumulate S (6 f64 instructions per Gauss point in 2-D)
//     into a wave-private LDS tile [entry][row][slot] with ds_add_f64; the map T = C[v(.,.)][v(.,.)] is applied once
//     per STORED block at the end (a kernel argument: scalar registers);
//   * the tile is the chunk's part of the slice in K's own layout: written as 16 rows x 16 bytes = 256 contiguous
//     bytes per (slot, entry pair), padding slots and padding rows as the zeros the tile started with.
// Deterministic: fixed pair order, ds_add_f64 of one instruction applied in lane order (asserted per box by
// test_fullsize_properties), no workgroup barrier at all.  Two steps of records are in flight (register sets A / B).
// Where its time goes (build with -DFEMCY_PAIRS_PROBE=<bits>, results meaningless; profiles/r06_pairs_probe.txt): bit 1 no
// global stores, 2 no LDS atomics, 4 no record loads, 8 no tile zeroing / LDS reads at the end, 16 no cross-lane reads.
#ifdef FEMCY_PAIRS_PROBE
#define PAIRS_PROBE_BIT(b_) ((FEMCY_PAIRS_PROBE & (b_)) != 0)
#else
#define PAIRS_PROBE_BIT(b_) 0
#endif
// the value of lane (quad base + G) in every lane of the quad: v_mov_b32 with quad_perm:[G, G, G, G], no LDS traffic
template <int G>
__device__ __forceinline__ double quad_bcast(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), G * 0x55, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), G * 0x55, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int DD>
struct SumMap {
    double t[DD * DD];      // K[ik] = sum_jl t[ik * DD + jl] S[jl]
};

// RPW = rows per wavefront (a chunk: 16 or 8 adjacent rows = 256 / 128 contiguous bytes per store run), DEPTH = steps of
// records in flight per wave, XCDC = workgroup b takes the (b / 8)-th unit of XCD (b % 8)'s CONTIGUOUS share of the
// processing order (workgroups are dispatched round-robin over the XCDs: an element's record is then fetched into ONE
// L2 and found there by the other rows of the element, instead of into up to eight).
//
// A wave takes a UNIT of consecutive chunks of the processing order and runs them as ONE stream of steps.  The host
// cuts every chunk into batches of <= 64 pairs (one register of pair codes; a CPE8 chunk is one batch) and a batch runs
// as groups of DEPTH steps; the loop below is one group per trip, straight-line, with ONE exit at the latch and every
// load unconditional -- the form in which the compiler counts its waits (s_waitcnt vmcnt(N), N = the loads issued
// since) instead of draining the queue: the first form of this kernel (a loop nest over chunks, batches and steps with
// conditional prefetches) compiled to vmcnt(0) before every compute and was as fast with 2 as with 4 steps in flight,
// with and without prefetch across chunks (profiles/r06_asm_cpe8_knobs.txt).  While group g computes, the records of
// group g + 1 are in flight -- in the last group of a batch the first group of the NEXT batch, whose pair codes were
// requested a batch earlier and whose descriptor two batches earlier (scalar loads): the chain descriptor -> codes ->
// records is paid once per wave.  The write-out leaves the tile zeroed (each LDS word is read and cleared by one lane).
constexpr int PAIR_ROW_SHIFT = 27;  // pair word: row inside the chunk << 27 | element * npe + local node (ensure_pairs)
struct PairBatch {
    int32_t chunk, p0, nb, L;       // chunk id (slice * (64 / RPW) + part), first pair, pairs (<= 64), slice width in blocks
    int64_t off;                    // slice_off of the chunk's slice
    int32_t last, pad;              // 1 = last batch of its chunk: write the tile out
};

template <int NPE, int NGP, int DM, int RPW, int DEPTH, bool XCDC>
__global__ void __launch_bounds__(256) k_assemble_pairs(int32_t nunits, int32_t Lmax,
                                                        const int32_t* __restrict__ unit_ptr,
                                                        const PairBatch* __restrict__ pr_desc,
                                                        const int32_t* __restrict__ pr_code,
                                                        const uint16_t* __restrict__ slotj,
                                                        const double* __restrict__ dsdx, const double* __restrict__ vol,
                                                        const SumMap<DM * DM> T, double* __restrict__ Kvals) {
    constexpr int DD = DM * DM, PPW = 64 / NPE, RD = NGP * NPE * DM, CPS = SLICE / RPW;
    static_assert(NGP <= NPE, "det J w of Gauss point g is fetched by the lane of column node g");
    extern __shared__ __attribute__((aligned(16))) double lds_pairs[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t wg = blockIdx.x;
    if (XCDC) {
        const int32_t per = gridDim.x >> 3;                      // the launch rounds the grid up to a multiple of 8
        wg = (wg & 7) * per + (wg >> 3);
    }
    const int32_t unit = wg * 4 + wave;
    if (unit >= nunits) return;
    int32_t b = __builtin_amdgcn_readfirstlane(unit_ptr[unit]);
    const int32_t bend = __builtin_amdgcn_readfirstlane(unit_ptr[unit + 1]);     // > b: a chunk has at least one batch
    double* __restrict__ acc = lds_pairs + (size_t)wave * (DD * RPW * (Lmax | 1));
    if (!PAIRS_PROBE_BIT(8))
        for (int i = lane; i < DD * RPW * (Lmax | 1); i += 64) acc[i] = 0.0;
    const int q = lane / NPE, t = lane - q * NPE, gbase = q * NPE;
    const bool lane_ok = q < PPW;                                // NPE = 6: lanes 60..63 have no pair
    const int tg = t % NGP;

    double G[DEPTH][NGP][DM];
    double V[DEPTH];
    int32_t J[DEPTH], C[DEPTH], R[DEPTH];
    // every load is unconditional: a lane without a pair reads pair 0 of the batch and drops the values, an empty batch
    // reads the zero padding behind the lists (element 0), the batch after the unit's last one is that batch again
    auto load = [&](int d, int st, int32_t codes, int32_t nb) {
        const int pi = st * PPW + q;
        const bool ok = lane_ok && pi < nb;
        const int32_t packed = __shfl(codes, ok ? pi : 0, 64);   // row of the chunk << 27 | element * NPE + local row node
        C[d] = packed & ((1 << PAIR_ROW_SHIFT) - 1);
        R[d] = packed >> PAIR_ROW_SHIFT;
        const int64_t e = C[d] / NPE;
        const double* __restrict__ rb = dsdx + e * RD + t * DM;
        if (PAIRS_PROBE_BIT(4)) {
#pragma unroll
            for (int g = 0; g < NGP; ++g)
#pragma unroll
                for (int i = 0; i < DM; ++i) G[d][g][i] = (double)(C[d] + g + i);
            V[d] = 1.0;
            J[d] = (C[d] + t) % 13;
            return;
        }
#pragma unroll
        for (int g = 0; g < NGP; ++g) load_row<DM>(rb + g * (NPE * DM), G[d][g]);
        V[d] = vol[e * NGP + tg];
        J[d] = slotj[(int64_t)C[d] * NPE + t];
    };
    auto compute = [&](int d, int st, int32_t nb, int Lp) {
        const int pi = st * PPW + q;
        const bool ok = lane_ok && pi < nb;
        const int la = C[d] % NPE;
        const int rl = R[d];
        double S[DD];
#pragma unroll
        for (int k = 0; k < DD; ++k) S[k] = 0.0;
#pragma unroll
        for (int g = 0; g < NGP; ++g) {
            double v;
            if constexpr (NGP == 4 && NPE % 4 == 0) {
                // lane t holds det J w of Gauss point t % 4: position g of EVERY quad -> a quad broadcast (DPP, no LDS)
                v = g == 0 ? quad_bcast<0>(V[d]) : g == 1 ? quad_bcast<1>(V[d]) : g == 2 ? quad_bcast<2>(V[d]) : quad_bcast<3>(V[d]);
            } else {
                v = PAIRS_PROBE_BIT(16) ? V[d] : __shfl(V[d], gbase + g, 64);
            }
#pragma unroll
            for (int i = 0; i < DM; ++i) {
                const double a = (PAIRS_PROBE_BIT(16) ? G[d][g][i] + la : __shfl(G[d][g][i], gbase + la, 64)) * v;
#pragma unroll
                for (int k = 0; k < DM; ++k) S[i * DM + k] += a * G[d][g][k];
            }
        }
        if (ok) {
            double* dst = acc + rl * Lp + J[d];
            if (PAIRS_PROBE_BIT(2)) {
                if (S[0] + S[1] + S[DD - 1] == 1.2345) dst[0] = S[0];
            } else {
#pragma unroll
                for (int k = 0; k < DD; ++k) atomicAdd(dst + k * (RPW * Lp), S[k]);
            }
        }
    };
    auto list_load = [&](const PairBatch& B, int32_t& codes) { codes = pr_code[B.p0 + (lane < B.nb ? lane : 0)]; };

    // the state advances by SELECTS, not branches: a load into a loop-carried register inside a conditional block makes
    // the compiler copy the freshly loaded register at the latch -- a wait for everything in flight, every trip
    auto sel = [](bool c, const PairBatch& x, const PairBatch& y) {
        PairBatch r;
        r.chunk = c ? x.chunk : y.chunk;
        r.p0 = c ? x.p0 : y.p0;
        r.nb = c ? x.nb : y.nb;
        r.L = c ? x.L : y.L;
        r.off = c ? x.off : y.off;
        r.last = c ? x.last : y.last;
        r.pad = 0;
        return r;
    };
    const int32_t blast = bend - 1;
    PairBatch B = pr_desc[b];
    PairBatch Bn = pr_desc[min(b + 1, blast)];
    PairBatch Bnn = pr_desc[min(b + 2, blast)];
    PairBatch Bn3 = pr_desc[min(b + 3, blast)];
    int32_t codes, codes_n, codes_nn;
    list_load(B, codes);
    list_load(Bn, codes_n);
    list_load(Bnn, codes_nn);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, d, codes, B.nb);
    int32_t g = 0;
    do {
        const int32_t nst = (B.nb + PPW - 1) / PPW;
        const bool last_group = (g + 1) * DEPTH >= nst;          // also the only group of an empty batch
        const int Lp = B.L | 1;                                  // odd row stride: the rows of a read start in different banks
        const int32_t src_codes = last_group ? codes_n : codes;
        const int32_t src_nb = last_group ? Bn.nb : B.nb;
        const int32_t src_st0 = last_group ? 0 : (g + 1) * DEPTH;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            compute(d, g * DEPTH + d, B.nb, Lp);
            load(d, src_st0 + d, src_codes, src_nb);
        }
        if (last_group && B.last) {
            wave_lds_sync();                                     // the atomics have landed
            // write-out: lane (rl, js) takes slots js, js + 64 / RPW, ... of row rl; RPW lanes = RPW adjacent rows =
            // 16 RPW contiguous bytes; the tile is left zeroed for the next chunk
            const int32_t s = B.chunk / CPS, r0 = (B.chunk - s * CPS) * RPW;
            const int rl = lane & (RPW - 1), js = lane / RPW;
            const int r = r0 + rl;
            double* __restrict__ Krow = Kvals + B.off * (int64_t)(DD * SLICE);
            for (int32_t j = js; j < B.L; j += 64 / RPW) {
                double S[DD], Kb[DD];
#pragma unroll
                for (int k = 0; k < DD; ++k) {
                    S[k] = PAIRS_PROBE_BIT(8) ? (double)(j + k) : acc[(k * RPW + rl) * Lp + j];
                    if (!PAIRS_PROBE_BIT(8)) acc[(k * RPW + rl) * Lp + j] = 0.0;
                }
#pragma unroll
                for (int ik = 0; ik < DD; ++ik) {
                    double a = 0.0;
#pragma unroll
                    for (int jl = 0; jl < DD; ++jl) a += T.t[ik * DD + jl] * S[jl];
                    Kb[ik] = a;
                }
                double* dst = Krow + (int64_t)j * (DD * SLICE);
                if (PAIRS_PROBE_BIT(1)) {
                    if (Kb[0] + Kb[1] + Kb[DD - 1] == 1.2345) dst[r] = Kb[0];
                    continue;
                }
#pragma unroll
                for (int pc = 0; pc < DD / 2; ++pc)
                    reinterpret_cast<double2*>(dst + pc * (2 * SLICE))[r] = make_double2(Kb[2 * pc], Kb[2 * pc + 1]);
                if (DD & 1) dst[(DD / 2) * (2 * SLICE) + r] = Kb[DD - 1];
            }
            wave_lds_sync();                                     // the zeros are in place before the next chunk's atomics
        }
        const bool adv = last_group;
        b = adv ? b + 1 : b;
        g = adv ? 0 : g + 1;
        B = sel(adv, Bn, B);
        codes = adv ? codes_n : codes;
        Bn = sel(adv, Bnn, Bn);
        codes_n = adv ? codes_nn : codes_n;                      // requested a trip ago or earlier: no wait beyond the records'
        Bnn = sel(adv, Bn3, Bnn);
        list_load(Bnn, codes_nn);                                // every trip (the same list again while the batch lasts)
        Bn3 = pr_desc[min(b + 3, blast)];
    } while (b < bend);
}

// ----------------------------------------------------------------------------- nodal force gather
// assemble_nodal_force_GN_kernel (stiffnessMtrx.py:620-644) is node-parallel with a serial loop over the padded
// nodeEles row that reads a gradient row, a stress tensor and a weight per (element, Gauss point).  Here the element
// pass (k_geom) leaves fe[e][a][:] = sum_g gradN_a . sigma * vol, half a wavefront (32 lanes) owns a node and the lanes
// are its incident elements: one dm-double load each, side by side instead of ~24 dependent load chains one after
// the other; the dm partial sums are combined by a fixed xor-shuffle tree (deterministic) and lane 0 stores.
template <int DM>
__global__ void __launch_bounds__(256) k_nodal_force(int32_t nn, const int32_t* __restrict__ ne_ptr,
                                                     const int32_t* __restrict__ ne_idx,
                                                     const double* __restrict__ fe, double* __restrict__ f) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t a = (int32_t)(t >> 5);
    const int sub = (int)(t & 31);
    double acc[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i) acc[i] = 0.0;
    if (a < nn) {
        const int32_t k1 = ne_ptr[a + 1];
        for (int32_t k = ne_ptr[a] + sub; k < k1; k += 32) {
            const double* __restrict__ row = fe + (int64_t)ne_idx[k] * DM;      // ne_idx = e*npe + la: the row of fe
            double rv[DM];
            load_row<DM>(row, rv);
#pragma unroll
            for (int i = 0; i < DM; ++i) acc[i] += rv[i];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < DM; ++i) acc[i] += __shfl_xor(acc[i], o, 32);
    if (a < nn && sub == 0) {
#pragma unroll
        for (int i = 0; i < DM; ++i) f[(int64_t)a * DM + i] = acc[i];
    }
}

// ------------------------------------------------------------------ Dirichlet 0/1 on the matrix
// thread (q, j): constrained scalar DOF q, slot j of its node's row.  Zero row r of block (a, b_j),
// zero column r of the mirror block (b_j, a), finally K[a][a][r][r] = 1 (only thread j == 0 touches it).
template <int DM>
__global__ void __launch_bounds__(256) k_dirichlet_zero(int32_t k, int32_t maxL, const int32_t* __restrict__ dofs,
                                                        const int64_t* __restrict__ slice_off,
                                                        const int32_t* __restrict__ rowlen,
                                                        const int32_t* __restrict__ pos,
                                                        const int32_t* __restrict__ bcol, double* __restrict__ Kvals,
                                                        double* __restrict__ resid,
                                                        const uint8_t* __restrict__ owner) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)k * maxL) return;
    const int32_t q = (int32_t)(t / maxL), j = (int32_t)(t % maxL);
    const int32_t dof = dofs[q];
    const int32_t a = dof / DM, r = dof % DM;
    if (j >= rowlen[a]) return;
    const int32_t pa = pos[a];
    const int64_t rowa = slice_off[pa >> 6] + j;
    const int lanea = pa & 63;
#pragma unroll
    for (int cc = 0; cc < DM; ++cc) Kvals[kv_index<DM>(rowa, r * DM + cc, lanea)] = 0.0;
    const int32_t b = bcol[rowa * SLICE + lanea];
    // mirror block: slot of a in the row of b (diagonal first, then ascending)
    int32_t js = 0;
    if (b != a) {
        const int32_t pb = pos[b];
        const int64_t offb = slice_off[pb >> 6];
        const int laneb = pb & 63;
        const int32_t Lb = rowlen[b];
        js = -1;
        for (int32_t jj = 1; jj < Lb; ++jj)
            if (bcol[(offb + jj) * SLICE + laneb] == a) js = jj;
    }
    if (js >= 0) {
        const int32_t pb = pos[b];
        const int64_t rowb = slice_off[pb >> 6] + js;
        const int laneb = pb & 63;
#pragma unroll
        for (int cc = 0; cc < DM; ++cc) Kvals[kv_index<DM>(rowb, cc * DM + r, laneb)] = 0.0;
    }
    if (j == 0) {
        // multi-rank: K is sub-assembled, so only the owning rank contributes the unit diagonal
        Kvals[kv_index<DM>(rowa, r * DM + r, lanea)] = owner ? (double)owner[dof] : 1.0;
        if (resid) resid[dof] = 0.0;
    }
}

// (cauchy_small, energy_density, post_point: element_math.hpp -- shared with the host backend)

template <int DM>
__global__ void __launch_bounds__(256) k_post(int64_t ngp, int large, int kind, const double* __restrict__ C, double p0,
                                              double p1, const double* __restrict__ Fin, double* __restrict__ sigma,
                                              double* __restrict__ strain, double* __restrict__ mises) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngp) return;
    post_point<DM>(large, kind, C, p0, p1, Fin + t * DM * DM, sigma + t * DM * DM, strain + t * DM * DM, mises + t);
}

template <int DM>
__global__ void __launch_bounds__(256) k_energy(int64_t ngp, int kind, const double* __restrict__ C, double p0, double p1,
                                                const double* __restrict__ Fin, double* __restrict__ energy) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngp) return;
    double F[DM][DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int j = 0; j < DM; ++j) F[i][j] = Fin[t * DM * DM + i * DM + j];
    energy[t] = energy_density<DM>(kind, C, p0, p1, F);
}

// ELE.extrapolate (element_zoo/*.py): nodal_vals[e][a] = sum_g E[a][g] * field[e][g][comp]
__global__ void __launch_bounds__(256) k_extrapolate(int64_t ne, int npe, int nGP, int width, int comp,
                                                     const double* __restrict__ E, const double* __restrict__ field,
                                                     double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne * npe) return;
    const int64_t e = t / npe;
    const int a = (int)(t % npe);
    double acc = 0.0;
    for (int g = 0; g < nGP; ++g) acc += E[a * nGP + g] * field[(e * nGP + g) * width + comp];
    out[t] = acc;
}

// sum_i a[i] * b[i] partials (elastic energy = sum density * vol)
__global__ void __launch_bounds__(256) k_dot_partial(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                     double* __restrict__ part) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += a[i] * b[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

int launch_post(Ctx* c, int large) {
    const int64_t ngp = (int64_t)c->ne * c->nGP;
    const int bs = 256, grid = (int)((ngp + bs - 1) / bs);
    if (c->dm == 3)
        hipLaunchKernelGGL((k_post<3>), dim3(grid), dim3(bs), 0, c->stream, ngp, large, c->mat_kind, c->d_C,
                           c->mat_params[0], c->mat_params[1], c->d_F, c->d_sigma, c->d_strain, c->d_mises);
    else
        hipLaunchKernelGGL((k_post<2>), dim3(grid), dim3(bs), 0, c->stream, ngp, large, c->mat_kind, c->d_C,
                           c->mat_params[0], c->mat_params[1], c->d_F, c->d_sigma, c->d_strain, c->d_mises);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int launch_energy(Ctx* c) {
    const int64_t ngp = (int64_t)c->ne * c->nGP;
    const int bs = 256, grid = (int)((ngp + bs - 1) / bs);
    if (c->dm == 3)
        hipLaunchKernelGGL((k_energy<3>), dim3(grid), dim3(bs), 0, c->stream, ngp, c->mat_kind, c->d_C,
                           c->mat_params[0], c->mat_params[1], c->d_F, c->d_energy);
    else
        hipLaunchKernelGGL((k_energy<2>), dim3(grid), dim3(bs), 0, c->stream, ngp, c->mat_kind, c->d_C,
                           c->mat_params[0], c->mat_params[1], c->d_F, c->d_energy);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------- Neumann loads
// neumannBC (stiffnessMtrx.py:369-411): one lane per loaded facet evaluates the facet's integration points on the
// UNDEFORMED element (dead load) and leaves the load of each facet node in contrib[facet][node][dm]; a second
// kernel sums the contributions of each loaded node in ascending (facet, node slot) order -- no atomics, the same
// bits on every run.  Loaded surfaces have 1e2..1e5 facets: launch-latency sized, kept on the device so that an
// increment never waits for a host loop over facets.
template <int DM>
__global__ void __launch_bounds__(128) k_neumann_contrib(int32_t nload, int32_t npe, int32_t nfn, int32_t nip,
                                                         const double* __restrict__ nodes,
                                                         const int32_t* __restrict__ elems,
                                                         const int32_t* __restrict__ load_elem,
                                                         const int32_t* __restrict__ load_ft,
                                                         const int32_t* __restrict__ ft_nodes,
                                                         const double* __restrict__ ft_N,
                                                         const double* __restrict__ ft_dN,
                                                         const double* __restrict__ ft_normal,
                                                         const double* __restrict__ ft_weight, double traction,
                                                         const double* __restrict__ dir_or_null,
                                                         double* __restrict__ contrib) {
    const int32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nload) return;
    neumann_facet<DM>(npe, nfn, nip, nodes, elems + (int64_t)load_elem[l] * npe, load_ft[l], ft_nodes, ft_N, ft_dN,
                      ft_normal, ft_weight, traction, dir_or_null, contrib + (int64_t)l * nfn * DM);
}

__global__ void __launch_bounds__(128) k_neumann_gather(int32_t nnode, int32_t dm, const int32_t* __restrict__ ld_node,
                                                        const int32_t* __restrict__ ld_ptr,
                                                        const int32_t* __restrict__ ld_slot,
                                                        const double* __restrict__ contrib, double* __restrict__ rhs) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnode) return;
    for (int d = 0; d < dm; ++d) {
        double s = 0.0;
        for (int32_t k = ld_ptr[i]; k < ld_ptr[i + 1]; ++k) s += contrib[(int64_t)ld_slot[k] * dm + d];
        rhs[(int64_t)ld_node[i] * dm + d] = s;
    }
}

int launch_neumann(Ctx* c, const Ctx::LoadSet& ls, double traction, bool along_normal, double* d_rhs) {
    FEMCY_HIP(hipMemsetAsync(d_rhs, 0, sizeof(double) * c->n, c->stream));
    if (ls.nload == 0) return FEMCY_OK;
    const int bs = 128;
    const double* dir = along_normal ? nullptr : ls.d_dir;
    if (c->dm == 3)
        hipLaunchKernelGGL((k_neumann_contrib<3>), dim3((ls.nload + bs - 1) / bs), dim3(bs), 0, c->stream, ls.nload,
                           c->npe, ls.nfn, ls.nip, c->d_nodes, c->d_elems, ls.d_elem, ls.d_ft, ls.d_ft_nodes, ls.d_N,
                           ls.d_dN, ls.d_normal, ls.d_weight, traction, dir, ls.d_contrib);
    else
        hipLaunchKernelGGL((k_neumann_contrib<2>), dim3((ls.nload + bs - 1) / bs), dim3(bs), 0, c->stream, ls.nload,
                           c->npe, ls.nfn, ls.nip, c->d_nodes, c->d_elems, ls.d_elem, ls.d_ft, ls.d_ft_nodes, ls.d_N,
                           ls.d_dN, ls.d_normal, ls.d_weight, traction, dir, ls.d_contrib);
    hipLaunchKernelGGL(k_neumann_gather, dim3((ls.nnode + bs - 1) / bs), dim3(bs), 0, c->stream, ls.nnode, c->dm,
                       ls.d_node, ls.d_ptr, ls.d_slot, ls.d_contrib, d_rhs);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int launch_extrapolate(Ctx* c, const double* d_E, const double* d_field, int width, int comp, double* d_out) {
    const int64_t total = (int64_t)c->ne * c->npe;
    hipLaunchKernelGGL(k_extrapolate, dim3((int)((total + 255) / 256)), dim3(256), 0, c->stream, (int64_t)c->ne, c->npe,
                       c->nGP, width, comp, d_E, d_field, d_out);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int launch_energy_sum(Ctx* c, double* total) {
    const int64_t ngp = (int64_t)c->ne * c->nGP;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((ngp + 255) / 256, 1024));
    hipLaunchKernelGGL(k_dot_partial, dim3(grid), dim3(256), 0, c->stream, ngp, c->d_energy, c->d_vol, c->d_part2);
    FEMCY_HIP(hipGetLastError());
    std::vector<double> h(grid);
    FEMCY_HIP(hipMemcpyAsync(h.data(), c->d_part2, sizeof(double) * grid, hipMemcpyDeviceToHost, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    double s = 0.0;
    for (double v : h) s += v;
    if (c->comm) {       // elements are held by exactly one rank: the total is the plain sum over the ranks
        FEMCY_HIP(hipMemcpyAsync(c->d_part2, &s, sizeof(double), hipMemcpyHostToDevice, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        return scalar_across_ranks(c, c->d_part2, 0, total);
    }
    *total = s;
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------- host launchers
#define FEMCY_DISPATCH_ELEMENT(NPE_, DM_, CALL)                          \
    if (c->npe == NPE_ && c->dm == DM_) {                                \
        constexpr int NPE = NPE_, DM = DM_;                              \
        CALL;                                                            \
        launched = true;                                                 \
    }

int launch_geom(Ctx* c, const double* d_u, unsigned what) {
    const int bs = 256, grid = (c->ne + bs - 1) / bs;
    bool launched = false;
    const bool with_stress = (what & (GEOM_F | GEOM_SIGMA | GEOM_FE)) != 0;
    size_t th = timing_begin(c, T_GEOM);
#define GEOM_CALL                                                                                                   \
    if (with_stress)                                                                                                \
        hipLaunchKernelGGL((k_geom<NPE, DM, true>), dim3(grid), dim3(bs), 0, c->stream, c->ne, c->nGP, c->d_nodes,  \
                           d_u, c->d_elems, c->d_dN, c->d_w, c->mat_kind, c->d_C, c->mat_params[0],                 \
                           c->mat_params[1], (what & GEOM_DSDX) ? c->d_dsdx : nullptr, c->d_vol,                    \
                           (what & GEOM_F) ? c->d_F : nullptr, (what & GEOM_SIGMA) ? c->d_sigma : nullptr,          \
                           (what & GEOM_FE) ? c->d_fe : nullptr);                                                   \
    else                                                                                                            \
        hipLaunchKernelGGL((k_geom<NPE, DM, false>), dim3(grid), dim3(bs), 0, c->stream, c->ne, c->nGP, c->d_nodes, \
                           d_u, c->d_elems, c->d_dN, c->d_w, c->mat_kind, c->d_C, c->mat_params[0],                 \
                           c->mat_params[1], c->d_dsdx, c->d_vol, c->d_F, c->d_sigma, (double*)nullptr)
    FEMCY_DISPATCH_ELEMENT(3, 2, GEOM_CALL)
    FEMCY_DISPATCH_ELEMENT(4, 2, GEOM_CALL)
    FEMCY_DISPATCH_ELEMENT(6, 2, GEOM_CALL)
    FEMCY_DISPATCH_ELEMENT(8, 2, GEOM_CALL)
    FEMCY_DISPATCH_ELEMENT(4, 3, GEOM_CALL)
    FEMCY_DISPATCH_ELEMENT(10, 3, GEOM_CALL)
#undef GEOM_CALL
    timing_end(c, th);
    if (!launched) {
        set_error("no geometry kernel instantiated for npe=%d dm=%d", c->npe, c->dm);
        return FEMCY_ENOKERNEL;
    }
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

// FEMCY_ASM_PAIRS: instantiated element families, LDS of a workgroup (four chunk tiles [dm^2][rows per wave][Lmax | 1])
static bool pairs_instantiated(const Ctx* c) {
    return c->dm == 2 && ((c->npe == 8 && c->nGP == 4) || (c->npe == 6 && c->nGP == 3) || (c->npe == 4 && c->nGP == 4) ||
                          (c->npe == 3 && c->nGP == 1));
}
static bool pairs_fits(const Ctx* c) { return (int64_t)c->ne * c->npe < ((int64_t)1 << 27); }   // the packed pair word
static int pairs_rpw(const Ctx* c) { return ((c->tune_pairs >> 1) & 3) == 1 ? 8 : 16; }
static size_t pairs_lds(const Ctx* c) {
    return (size_t)4 * c->dm * c->dm * pairs_rpw(c) * (c->max_row_blocks | 1) * sizeof(double);
}
// T[(i,k)][(j,l)] = C[v(i,j)][v(k,l)], v = the Voigt index of the reference's B matrices (kblock_add)
template <int DM>
static SumMap<DM * DM> sum_map(const Ctx* c) {
    constexpr int NS = DM == 3 ? 6 : 3;
    auto v = [](int i, int j) { return i == j ? i : (DM == 2 ? 2 : i + j + 2); };   // 3-D: (0,1) 3, (0,2) 4, (1,2) 5
    SumMap<DM * DM> m;
    for (int i = 0; i < DM; ++i)
        for (int k = 0; k < DM; ++k)
            for (int j = 0; j < DM; ++j)
                for (int l = 0; l < DM; ++l) m.t[(i * DM + k) * DM * DM + j * DM + l] = c->h_C[v(i, j) * NS + v(k, l)];
    return m;
}

int launch_assemble(Ctx* c) {
    const int bs = 256;
    size_t th = timing_begin(c, T_ASM);
    int mode = c->opt_assembly;
    if (mode == FEMCY_ASM_AUTO) {
        mode = (c->npe > 4) ? ((c->dm == 3 && c->npe == 10 && c->nGP == 4 && c->dN_sums_to_zero) ? FEMCY_ASM_ROWS2 : FEMCY_ASM_ROWS)
                            : (c->dN_sums_to_zero ? FEMCY_ASM_GATHER_SYM_ROWSUM : FEMCY_ASM_GATHER_SYM);
        // C3D10: two rows per wave (round 3: 386 -> 307 us on the bench mesh) when its accumulators fit the LDS; it computes
        // the diagonal block like the others, so it does not need element tables whose gradients sum to zero
        if (c->dm == 3 && c->npe == 10 && c->nGP == 4) {
            const size_t lds4 = (size_t)4 * 2 * (3 * 120 + 12 + ((c->max_row_blocks * 9 + 1) & ~1) + 2) * sizeof(double);
            if (lds4 + 512 <= (size_t)c->small_max_lds) mode = FEMCY_ASM_ROWS4;
        }
        // round 6: the 2-D quadratic families (many short rows) -- 16 rows per wave, pair lists in storage order
        if (c->dm == 2 && c->npe > 4 && pairs_instantiated(c) && pairs_fits(c) && pairs_lds(c) + 512 <= (size_t)c->small_max_lds)
            mode = FEMCY_ASM_PAIRS;
    }
    if (c->opt_tangent == 1) {
        FEMCY_REQUIRE(c->mat_kind != FEMCY_MAT_PSTRESS, "the consistent tangent is not available for plane stress");
        const bool neo = c->mat_kind == FEMCY_MAT_NEOHOOKE;
        const int s = c->dm == 3 ? 6 : 3;
        const double lam = c->h_C[0 * s + 1], mu = c->h_C[(s - 1) * s + (s - 1)];     // isotropic C: C01, C(shear,shear)
        const int64_t npos = c->stored_rows * SLICE;
        const int grid = (int)((npos + bs - 1) / bs);
        const int skip = c->dN_sums_to_zero ? 1 : 0;
        if (c->dm == 3)
            hipLaunchKernelGGL((k_assemble_gather_consistent<3>), dim3(grid), dim3(bs), 0, c->stream, npos, c->npe, c->nGP,
                               c->d_ctr_ptr, c->d_ctr, c->d_dsdx, c->d_vol, c->d_F, c->d_sigma, neo, lam, mu,
                               c->mat_params[0], c->mat_params[1], c->d_tpos, skip, c->d_Kvals);
        else
            hipLaunchKernelGGL((k_assemble_gather_consistent<2>), dim3(grid), dim3(bs), 0, c->stream, npos, c->npe, c->nGP,
                               c->d_ctr_ptr, c->d_ctr, c->d_dsdx, c->d_vol, c->d_F, c->d_sigma, neo, lam, mu,
                               c->mat_params[0], c->mat_params[1], c->d_tpos, skip, c->d_Kvals);
        if (skip) {
            const int64_t nposd = (int64_t)c->nslices * SLICE;
            const int gd = (int)((nposd + bs - 1) / bs);
            if (c->dm == 3)
                hipLaunchKernelGGL((k_diag_from_rowsum<3>), dim3(gd), dim3(bs), 0, c->stream, c->nslices, c->d_node_of,
                                   c->d_rowlen, c->d_slice_off, c->d_Kvals);
            else
                hipLaunchKernelGGL((k_diag_from_rowsum<2>), dim3(gd), dim3(bs), 0, c->stream, c->nslices, c->d_node_of,
                                   c->d_rowlen, c->d_slice_off, c->d_Kvals);
        }
        timing_end(c, th);
        FEMCY_HIP(hipGetLastError());
        return FEMCY_OK;
    }
    if (mode == FEMCY_ASM_ROWS3) {
        FEMCY_REQUIRE(c->dm == 3 && c->dN_sums_to_zero && ((c->npe == 10 && c->nGP == 4) || (c->npe == 4 && c->nGP == 1)),
                      "ROWS3 assembly is instantiated for C3D10 / C3D4 tables with sum_a dN_a = 0 (npe %d, nGP %d)", c->npe, c->nGP);
        const int T = c->npe - 1, EPC = 64 / T, RD = c->nGP * c->npe * 3;
        const int volw = (EPC * c->nGP + 1) & ~1, codew = (EPC + 1) / 2 * 2 / 2 + 1, accw = (c->max_row_blocks * 9 + 1) & ~1;
        const size_t lds = (size_t)4 * (EPC * RD + volw + 2 * accw + 2 * codew) * sizeof(double);
        if (lds + 512 > (size_t)c->small_max_lds) {
            FEMCY_REQUIRE(c->opt_assembly == FEMCY_ASM_AUTO, "ROWS3 assembly needs %zu B of LDS per workgroup (longest row: %d "
                          "blocks), the device allows %d", lds, c->max_row_blocks, c->small_max_lds);
            mode = FEMCY_ASM_ROWS2;
        }
#define FEMCY_ROWS3(NPE_, NGP_, CUB_)                                                                                  \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            FEMCY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows3<NPE_, NGP_, CUB_>),          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((k_assemble_rows3<NPE_, NGP_, CUB_>), dim3(c->nslices), dim3(bs), lds, c->stream, c->nslices, \
                           c->nn, c->max_row_blocks, c->d_ne_ptr, c->d_ne_idx, c->d_slotj, c->d_rowlen, c->d_node_of,  \
                           c->d_slice_off, c->d_dsdx, c->d_vol, c->d_C, c->cubic[0], c->cubic[1], c->cubic[2],         \
                           c->d_Kvals);                                                                                \
    } while (0)
        if (mode == FEMCY_ASM_ROWS3) {
            if (c->npe == 10) { if (c->C_is_cubic) FEMCY_ROWS3(10, 4, true); else FEMCY_ROWS3(10, 4, false); }
            else              { if (c->C_is_cubic) FEMCY_ROWS3(4, 1, true); else FEMCY_ROWS3(4, 1, false); }
        }
#undef FEMCY_ROWS3
    }
    if (mode == FEMCY_ASM_ROWS2) {
        FEMCY_REQUIRE(c->dm == 3 && c->dN_sums_to_zero && ((c->npe == 10 && c->nGP == 4) || (c->npe == 4 && c->nGP == 1)),
                      "ROWS2 assembly is instantiated for C3D10 / C3D4 tables with sum_a dN_a = 0 (npe %d, nGP %d)", c->npe, c->nGP);
        const int T = c->npe - 1, EPC = 64 / T, RD = c->nGP * c->npe * 3;
        const int volw = (EPC * c->nGP + 1) & ~1, codew = (EPC + 1) / 2 * 2 / 2 + 1, accw = (c->max_row_blocks * 9 + 1) & ~1;
#ifdef FEMCY_ROWS2_LDS_PAD      /* occupancy experiments: fewer workgroups per CU (profiles/r03_rows2_probe.txt) */
        const size_t lds = (size_t)4 * (EPC * RD + volw + accw + 2 * codew) * sizeof(double) + FEMCY_ROWS2_LDS_PAD;
#else
        const size_t lds = (size_t)4 * (EPC * RD + volw + accw + 2 * codew) * sizeof(double);
#endif
        // the accumulator of a row grows with the longest row of the mesh (288 B per block): an unstructured mesh with
        // high-valence nodes can exceed what a workgroup may allocate -- AUTO then takes ROWS (its LDS is 4 rows only)
        if (lds + 512 > (size_t)c->small_max_lds) {
            FEMCY_REQUIRE(c->opt_assembly == FEMCY_ASM_AUTO, "ROWS2 assembly needs %zu B of LDS per workgroup (longest row: %d "
                          "blocks), the device allows %d", lds, c->max_row_blocks, c->small_max_lds);
            mode = FEMCY_ASM_ROWS;
        }
#define FEMCY_ROWS2(NPE_, NGP_, CUB_)                                                                                  \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            FEMCY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows2<NPE_, NGP_, CUB_>),          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((k_assemble_rows2<NPE_, NGP_, CUB_>), dim3(c->nslices), dim3(bs), lds, c->stream, c->nslices, \
                           c->max_row_blocks, c->d_ne_ptr, c->d_ne_idx, c->d_slotj, c->d_rowlen, c->d_node_of,         \
                           c->d_slice_off, c->d_dsdx, c->d_vol, c->d_C, c->cubic[0], c->cubic[1], c->cubic[2],         \
                           c->d_Kvals);                                                                                \
    } while (0)
        if (mode == FEMCY_ASM_ROWS2) {
            if (c->npe == 10) { if (c->C_is_cubic) FEMCY_ROWS2(10, 4, true); else FEMCY_ROWS2(10, 4, false); }
            else              { if (c->C_is_cubic) FEMCY_ROWS2(4, 1, true); else FEMCY_ROWS2(4, 1, false); }
        }
#undef FEMCY_ROWS2
    }
    if (mode == FEMCY_ASM_ROWS4) {
        FEMCY_REQUIRE(c->dm == 3 && c->npe == 10 && c->nGP == 4, "ROWS4 assembly is instantiated for C3D10 (npe %d, nGP %d)",
                      c->npe, c->nGP);
        const int EPG = 32 / c->npe, RD = c->nGP * c->npe * 3;
#ifdef FEMCY_ROWS4_FAKE_LMAX    /* occupancy experiment only: WRONG results for longer rows */
        const int R4_LMAX = FEMCY_ROWS4_FAKE_LMAX;
#else
        const int R4_LMAX = c->max_row_blocks;
#endif
        const int volw = (EPG * c->nGP + 1) & ~1, accw = (R4_LMAX * 9 + 1) & ~1;
        // experiment (round 5): FEMCY_TUNE_ROWS4_TILE = 1000 GP + LCUT -- whole-line write-out from a wave's own LDS tile,
        // see the kernel; 0 (default) = off
        const int r4_gp = c->tune_rows4_tile / 1000, r4_lcut = c->tune_rows4_tile % 1000;
        const int wpair = 2 * (EPG * RD + volw + accw + 2);
        const int wtile = r4_gp ? 2 * (EPG * RD + volw + 2) + 2 * r4_gp * ((r4_lcut * 9) | 1) + 1 : 0;
        const int wstride = (std::max(wpair, wtile) + 1) & ~1;
        const size_t lds = (size_t)4 * wstride * sizeof(double);
        FEMCY_REQUIRE(lds + 512 <= (size_t)c->small_max_lds, "ROWS4 assembly needs %zu B of LDS per workgroup (longest row: %d "
                      "blocks), the device allows %d", lds, c->max_row_blocks, c->small_max_lds);
        // launch order: by decreasing work.  The locality order (FEMCY_TUNE_ROWS4_ORDER = 1) was built for meshes whose
        // records exceed the Infinity Cache and MEASURED slower at every size (124 k / 295 k / 995 k C3D10: 301 -> 354,
        // 754 -> 852, 2 477 -> 2 679 us, profiles/r06_rows4_order.txt) although it removes the re-fetches: this kernel is
        // bound by its LDS / request rate and by the balance of long and short slices, not by HBM -- kept as a knob
        const bool r4_near = c->tune_rows4_order == 1;
        const bool r4_natural = c->tune_rows4_order >= 2;      // experiments: 2 = storage order in XCD-contiguous ranges, 3 = storage order round-robin
        const int r4_grid = (r4_near || c->tune_rows4_order == 2) ? (c->nslices + 7) / 8 * 8 : c->nslices;
#define FEMCY_ROWS4(CUB_, GP_)                                                                                         \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            FEMCY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_rows4<10, 4, CUB_, GP_>),          \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((k_assemble_rows4<10, 4, CUB_, GP_>), dim3(r4_grid), dim3(bs), lds, c->stream, c->nslices,  \
                           R4_LMAX, r4_lcut, wstride, (r4_near || c->tune_rows4_order == 2) ? 1 : 0,                   \
                           (const int32_t*)(r4_natural ? c->d_asm_order_id : (r4_near ? c->d_asm_order_near : c->d_asm_order)), \
                           c->d_ne_ptr, c->d_ne_idx,                                                                   \
                           c->d_slotj, c->d_rowlen, c->d_node_of,                                                      \
                           c->d_slice_off, c->d_dsdx, c->d_vol, c->d_C, c->cubic[0], c->cubic[1], c->cubic[2],         \
                           c->d_Kvals);                                                                                \
    } while (0)
        if (r4_gp == 2) { if (c->C_is_cubic) FEMCY_ROWS4(true, 2); else FEMCY_ROWS4(false, 2); }
        else if (r4_gp == 4) { if (c->C_is_cubic) FEMCY_ROWS4(true, 4); else FEMCY_ROWS4(false, 4); }
        else { if (c->C_is_cubic) FEMCY_ROWS4(true, 0); else FEMCY_ROWS4(false, 0); }
#undef FEMCY_ROWS4
    } else if (mode == FEMCY_ASM_ROWS2 || mode == FEMCY_ASM_ROWS3) {
    } else if (mode == FEMCY_ASM_PAIRS) {
        FEMCY_REQUIRE(pairs_instantiated(c), "PAIRS assembly is instantiated for the 2-D families (npe %d, nGP %d, dm %d)",
                      c->npe, c->nGP, c->dm);
        FEMCY_REQUIRE(pairs_fits(c), "PAIRS assembly packs (row, element, local node) into 32 bits: ne * npe must stay below 2^27");
        // FEMCY_TUNE_PAIRS: bit 0 = XCD-contiguous ranges of the processing order, bits 1-2 = rows per wave (0: 16, 1: 8),
        // bits 3-4 = steps of records in flight (0: 2, 1: 3, 2: 4), bit 5 = chunks in Morton order of their centroids,
        // bits 6-9 = chunks per wave - 1
        const int tp = c->tune_pairs;
        const int cpw = 1 + ((tp >> 6) & 15);
        const int rpw = pairs_rpw(c), depth = 2 + ((tp >> 3) & 3);
        const bool xcdc = (tp & 1) != 0;
        const size_t lds = pairs_lds(c);
        FEMCY_REQUIRE(lds + 512 <= (size_t)c->small_max_lds, "PAIRS assembly needs %zu B of LDS per workgroup (longest row: %d "
                      "blocks), the device allows %d", lds, c->max_row_blocks, c->small_max_lds);
        int rc = ensure_pairs(c, rpw, (tp & 32) != 0, cpw);
        if (rc) return rc;
        const SumMap<4> T = sum_map<2>(c);
        const int32_t nchunks = c->nslices * (SLICE / rpw);
        const int32_t nunits = (nchunks + cpw - 1) / cpw;
        const int grid = ((nunits + 3) / 4 + 7) / 8 * 8;
#define FEMCY_PAIRS_K(NPE_, NGP_, RPW_, DEPTH_, X_)                                                                    \
    do {                                                                                                               \
        if (lds > 48 * 1024)                                                                                           \
            FEMCY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_assemble_pairs<NPE_, NGP_, 2, RPW_, DEPTH_, X_>), \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((k_assemble_pairs<NPE_, NGP_, 2, RPW_, DEPTH_, X_>), dim3(grid), dim3(bs), lds, c->stream,  \
                           nunits, c->max_row_blocks, c->d_pr_unit, (const PairBatch*)c->d_pr_ptr, c->d_pr_code,            \
                           c->d_slotj, c->d_dsdx, c->d_vol, T, c->d_Kvals);                                      \
    } while (0)
#define FEMCY_PAIRS_D(NPE_, NGP_, RPW_, X_)                                                                            \
    do {                                                                                                               \
        if (depth == 2) FEMCY_PAIRS_K(NPE_, NGP_, RPW_, 2, X_);                                                        \
        else if (depth == 3) FEMCY_PAIRS_K(NPE_, NGP_, RPW_, 3, X_);                                                   \
        else FEMCY_PAIRS_K(NPE_, NGP_, RPW_, 4, X_);                                                                   \
    } while (0)
#define FEMCY_PAIRS(NPE_, NGP_)                                                                                        \
    do {                                                                                                               \
        if (rpw == 16) { if (xcdc) FEMCY_PAIRS_D(NPE_, NGP_, 16, true); else FEMCY_PAIRS_D(NPE_, NGP_, 16, false); }   \
        else           { if (xcdc) FEMCY_PAIRS_D(NPE_, NGP_, 8, true); else FEMCY_PAIRS_D(NPE_, NGP_, 8, false); }     \
    } while (0)
        if (c->npe == 8) FEMCY_PAIRS(8, 4);
        else if (c->npe == 6) FEMCY_PAIRS(6, 3);
        else if (c->npe == 4) FEMCY_PAIRS(4, 4);
        else FEMCY_PAIRS(3, 1);
#undef FEMCY_PAIRS_K
#undef FEMCY_PAIRS_D
#undef FEMCY_PAIRS
    } else if (mode == FEMCY_ASM_ROWS) {
        const int grid = std::min((c->nn + 3) / 4, 256 * 16);
        const size_t lds = (size_t)4 * c->max_row_blocks * c->dm * c->dm * sizeof(double);
        FEMCY_REQUIRE(lds + 512 <= (size_t)c->small_max_lds, "ROWS assembly: a row of %d blocks does not fit the LDS",
                      c->max_row_blocks);
        if (lds > 48 * 1024)
            FEMCY_HIP(hipFuncSetAttribute(c->dm == 3 ? reinterpret_cast<const void*>(&k_assemble_rows<3>)
                                                     : reinterpret_cast<const void*>(&k_assemble_rows<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (c->dm == 3)
            hipLaunchKernelGGL((k_assemble_rows<3>), dim3(grid), dim3(bs), lds, c->stream, c->nn, c->npe, c->nGP,
                               c->max_row_blocks, c->d_ne_ptr, c->d_ne_idx, c->d_slotj, c->d_rowlen, c->d_pos, c->d_slice_off,
                               c->d_dsdx, c->d_vol, c->d_C, c->d_Kvals);
        else
            hipLaunchKernelGGL((k_assemble_rows<2>), dim3(grid), dim3(bs), lds, c->stream, c->nn, c->npe, c->nGP,
                               c->max_row_blocks, c->d_ne_ptr, c->d_ne_idx, c->d_slotj, c->d_rowlen, c->d_pos, c->d_slice_off,
                               c->d_dsdx, c->d_vol, c->d_C, c->d_Kvals);
    } else if (mode == FEMCY_ASM_ATOMIC) {
        FEMCY_HIP(hipMemsetAsync(c->d_Kvals, 0, (size_t)c->stored_rows * c->dm * c->dm * SLICE * sizeof(double),
                                 c->stream));
        const int64_t npair = (int64_t)c->ne * c->npe * c->npe;
        const int grid = (int)((npair + bs - 1) / bs);
        if (c->dm == 3)
            hipLaunchKernelGGL((k_assemble_atomic<3>), dim3(grid), dim3(bs), 0, c->stream, npair, c->npe, c->nGP,
                               c->d_elems, c->d_slotj, c->d_pos, c->d_slice_off, c->d_dsdx, c->d_vol, c->d_C, c->d_Kvals);
        else
            hipLaunchKernelGGL((k_assemble_atomic<2>), dim3(grid), dim3(bs), 0, c->stream, npair, c->npe, c->nGP,
                               c->d_elems, c->d_slotj, c->d_pos, c->d_slice_off, c->d_dsdx, c->d_vol, c->d_C, c->d_Kvals);
    } else {
        const int64_t npos = c->stored_rows * SLICE;
        const int grid = (int)((npos + bs - 1) / bs);
        const bool rowsum = mode == FEMCY_ASM_GATHER_SYM_ROWSUM;
        FEMCY_REQUIRE(!rowsum || c->dN_sums_to_zero, "row-sum diagonal needs element tables with sum_a dN_a = 0");
#define FEMCY_GATHER(DM_, SYM_, CUB_)                                                                                  \
    hipLaunchKernelGGL((k_assemble_gather<DM_, SYM_, CUB_>), dim3(grid), dim3(bs), 0, c->stream, npos, c->npe, c->nGP, \
                       c->d_ctr_ptr, c->d_ctr, c->d_tpos, c->d_dsdx, c->d_vol, c->d_C, c->cubic[0], c->cubic[1],        \
                       c->cubic[2], c->d_Kvals, rowsum ? 1 : 0)
        const bool sym = mode == FEMCY_ASM_GATHER_SYM || rowsum;
        if (c->dm == 3 && c->C_is_cubic) { if (sym) FEMCY_GATHER(3, true, true); else FEMCY_GATHER(3, false, true); }
        else if (c->dm == 3)             { if (sym) FEMCY_GATHER(3, true, false); else FEMCY_GATHER(3, false, false); }
        else                             { if (sym) FEMCY_GATHER(2, true, false); else FEMCY_GATHER(2, false, false); }
#undef FEMCY_GATHER
        if (rowsum) {
            const int64_t nposd = (int64_t)c->nslices * SLICE;
            const int gd = (int)((nposd + bs - 1) / bs);
            if (c->dm == 3)
                hipLaunchKernelGGL((k_diag_from_rowsum<3>), dim3(gd), dim3(bs), 0, c->stream, c->nslices, c->d_node_of,
                                   c->d_rowlen, c->d_slice_off, c->d_Kvals);
            else
                hipLaunchKernelGGL((k_diag_from_rowsum<2>), dim3(gd), dim3(bs), 0, c->stream, c->nslices, c->d_node_of,
                                   c->d_rowlen, c->d_slice_off, c->d_Kvals);
        }
    }
    timing_end(c, th);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int launch_nodal_force(Ctx* c, double* d_f) {
    const int bs = 256;
    const int grid = (int)(((int64_t)c->nn * 32 + bs - 1) / bs);   // 32 lanes per node
    size_t th = timing_begin(c, T_FORCE);
    if (c->dm == 3)
        hipLaunchKernelGGL((k_nodal_force<3>), dim3(grid), dim3(bs), 0, c->stream, c->nn, c->d_ne_ptr, c->d_ne_idx,
                           c->d_fe, d_f);
    else
        hipLaunchKernelGGL((k_nodal_force<2>), dim3(grid), dim3(bs), 0, c->stream, c->nn, c->d_ne_ptr, c->d_ne_idx,
                           c->d_fe, d_f);
    timing_end(c, th);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int launch_dirichlet_zero(Ctx* c, const int32_t* d_dofs, int32_t k, double* d_resid) {
    if (k <= 0) return FEMCY_OK;
    const int bs = 256;
    const int64_t total = (int64_t)k * c->max_row_blocks;
    const int grid = (int)((total + bs - 1) / bs);
    const uint8_t* owner = c->comm ? c->d_owner : nullptr;
    if (c->dm == 3)
        hipLaunchKernelGGL((k_dirichlet_zero<3>), dim3(grid), dim3(bs), 0, c->stream, k, c->max_row_blocks, d_dofs,
                           c->d_slice_off, c->d_rowlen, c->d_pos, c->d_bcol, c->d_Kvals, d_resid, owner);
    else
        hipLaunchKernelGGL((k_dirichlet_zero<2>), dim3(grid), dim3(bs), 0, c->stream, k, c->max_row_blocks, d_dofs,
                           c->d_slice_off, c->d_rowlen, c->d_pos, c->d_bcol, c->d_Kvals, d_resid, owner);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

}  // namespace femcy
