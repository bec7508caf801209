// libfemcy_cpu.so: the C ABI of include/femcy.h on the HOST (C++17 + OpenMP) -- SURVEY.md 8b "a CPU implementation of the
// same ABI", BASELINE configs[0] "plumbing, no GPU".  The reference runs on the CPU by one line (main.py:11) and solves
// small systems on the host (stiffnessMtrx.py:219-251); this backend lets `python -m femcy_amd.main deck.inp` and the
// whole Python surface run on a box without a GPU.
//
// It is NOT the oracle (`oracle/` restates the reference as written and is test infrastructure) and it is never chosen
// silently: femcy_amd.backend loads it only when FEMCY_BACKEND=cpu is set.  The arithmetic of an element, a Gauss
// point, a facet is the SAME CODE the HIP kernels run (csrc/element_math.hpp); what differs is storage and schedule:
//   * K is block-CSR (dm x dm blocks, the diagonal block first, then ascending columns -- the slot order of the
//     device's SELL matrix, so the reference-layout exports agree entry for entry), rows contiguous in memory;
//   * assembly is owner-computes by matrix row (one OpenMP task per node, its incident elements in ascending order:
//     no atomics, bit-reproducible for any thread count);
//   * reductions are sums of fixed chunks (1024 entries / 128 matrix rows) combined in order: the same bits for any thread count;
//   * PCG is the reference recurrence (conjugateGradientSolver.py:103-127) with the device's conventions (r0 = 0 ->
//     0 iterations, NaN/Inf -> FEMCY_ENUMERIC), fused as far as the dependencies allow: SpMV + d.Ad in one pass,
//     x / r update + r.M.r + max|r| in one pass, d update in one pass.
// Multi-rank entry points return FEMCY_ECOMM (one process = the whole mesh); the device probes return FEMCY_EINVAL.
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/femcy.h"
#include "../csrc/band_order.hpp"
#include "../csrc/element_math.hpp"

using namespace femcy;

namespace {

thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define REQUIRE(cond, ...)          \
    do {                            \
        if (!(cond)) {              \
            set_error(__VA_ARGS__); \
            return FEMCY_EINVAL;    \
        }                           \
    } while (0)

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int64_t CHUNK = 1024;   // reduction granule (vector entries): partial sums of fixed chunks, combined in order
constexpr int64_t NODE_CHUNK = 128;   // the same for loops over matrix rows (nodes)

struct DofSet {
    std::vector<int32_t> dofs;
};
struct LoadSet {
    int32_t nft, nfn, nip, nload;
    std::vector<int32_t> ft_nodes, elem, ft, node, ptr, slot;
    std::vector<double> N, dN, normal, weight, contrib;
};

}  // namespace

struct femcy_ctx {
    int32_t nn = 0, dm = 0, ne = 0, npe = 0, nGP = 0, s = 0;
    int64_t n = 0;
    std::vector<double> nodes, dN, w;
    std::vector<int32_t> elems;
    double C[36] = {0}, params[4] = {0, 0, 0, 0}, cubic[3] = {0, 0, 0};
    int32_t mat_kind = -1;
    bool C_is_cubic = false, have_mesh = false, have_element = false, have_material = false, have_pattern = false;
    // block-CSR matrix: row a = blocks rowptr[a] .. rowptr[a+1], diagonal first, then ascending columns
    std::vector<int64_t> rowptr;
    std::vector<int32_t> col;
    std::vector<double> K;
    std::vector<int64_t> eslot;          // [ne][npe][npe] element-local (a, b) -> block index
    std::vector<int32_t> ne_ptr, ne_idx; // node -> (element * npe + local index), ascending
    int32_t max_row_blocks = 0, max_node_elems = 0;
    // Gauss-point fields
    std::vector<double> dsdx, vol, F, sigma, strain, mises, energy, fe;
    std::vector<double> vec[FEMCY_VEC_COUNT], r, d, M, Ad;
    std::vector<DofSet> dofsets;
    std::vector<LoadSet> loadsets;
    int opt_tangent = 0, opt_timing = 0;
    femcy_timing_t timing{};
    // femcy_direct_solve: band order of the current pattern (built on first use), storage limit
    BandOrder band_order;
    bool have_band_order = false;
    int64_t direct_max_bytes = (int64_t)48 << 30;
};

namespace {

enum : unsigned { GEOM_DSDX = 1, GEOM_F = 2, GEOM_SIGMA = 4, GEOM_FE = 8 };

// get_dsdx_and_vol (stiffnessMtrx.py:132-150), get_deformation_gradient (:532-556), constitutiveOfLargeDeform and the
// per-element nodal forces of assemble_nodal_force_GN (:620-644): the element pass of the device's k_geom
template <int DM>
void geom_pass(femcy_ctx* c, const double* u, unsigned what) {
    const int npe = c->npe, nGP = c->nGP;
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < c->ne; ++e) {
        double X[27][DM], U[27][DM], facc[27][DM];
        for (int a = 0; a < npe; ++a) {
            const int32_t nd = c->elems[(int64_t)e * npe + a];
            for (int i = 0; i < DM; ++i) {
                X[a][i] = c->nodes[(int64_t)nd * DM + i];
                U[a][i] = u ? u[(int64_t)nd * DM + i] : 0.0;
                facc[a][i] = 0.0;
            }
        }
        for (int g = 0; g < nGP; ++g) {
            const double* dNg = c->dN.data() + (size_t)g * npe * DM;
            double J[DM][DM], inv[DM][DM];
            for (int i = 0; i < DM; ++i)
                for (int j = 0; j < DM; ++j) {
                    double acc = 0.0;
                    for (int a = 0; a < npe; ++a) acc += (X[a][i] + U[a][i]) * dNg[a * DM + j];
                    J[i][j] = acc;
                }
            const double det = det_inv<DM>(J, inv);
            const int64_t gp = (int64_t)e * nGP + g;
            if (what & GEOM_DSDX) {
                double* out = c->dsdx.data() + gp * npe * DM;
                for (int a = 0; a < npe; ++a)
                    for (int j = 0; j < DM; ++j) {
                        double acc = 0.0;
                        for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv[k][j];
                        out[a * DM + j] = acc;
                    }
                c->vol[gp] = det * c->w[g];
            }
            if (what & (GEOM_F | GEOM_SIGMA | GEOM_FE)) {
                double J0[DM][DM], inv0[DM][DM], F[DM][DM], sig[DM][DM];
                for (int i = 0; i < DM; ++i)
                    for (int j = 0; j < DM; ++j) {
                        double acc = 0.0;
                        for (int a = 0; a < npe; ++a) acc += X[a][i] * dNg[a * DM + j];
                        J0[i][j] = acc;
                        F[i][j] = 0.0;
                    }
                det_inv<DM>(J0, inv0);
                for (int a = 0; a < npe; ++a) {
                    double dsdX[DM];
                    for (int j = 0; j < DM; ++j) {
                        double acc = 0.0;
                        for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv0[k][j];
                        dsdX[j] = acc;
                    }
                    for (int i = 0; i < DM; ++i)
                        for (int j = 0; j < DM; ++j) F[i][j] += U[a][i] * dsdX[j];
                }
                for (int i = 0; i < DM; ++i) F[i][i] += 1.0;
                if (what & GEOM_F)
                    for (int i = 0; i < DM; ++i)
                        for (int j = 0; j < DM; ++j) c->F[gp * DM * DM + i * DM + j] = F[i][j];
                if (what & (GEOM_SIGMA | GEOM_FE)) {
                    cauchy_large<DM>(c->mat_kind, c->C, c->params[0], c->params[1], F, sig);
                    if (what & GEOM_SIGMA)
                        for (int i = 0; i < DM; ++i)
                            for (int j = 0; j < DM; ++j) c->sigma[gp * DM * DM + i * DM + j] = sig[i][j];
                    if (what & GEOM_FE) {
                        const double vg = det * c->w[g];
                        for (int a = 0; a < npe; ++a) {
                            double ga[DM];
                            for (int j = 0; j < DM; ++j) {
                                double acc = 0.0;
                                for (int k = 0; k < DM; ++k) acc += dNg[a * DM + k] * inv[k][j];
                                ga[j] = acc;
                            }
                            for (int i = 0; i < DM; ++i) {
                                double dsum = 0.0;
                                for (int j = 0; j < DM; ++j) dsum += ga[j] * sig[j][i];
                                facc[a][i] += dsum * vg;
                            }
                        }
                    }
                }
            }
        }
        if (what & GEOM_FE)
            for (int a = 0; a < npe; ++a)
                for (int i = 0; i < DM; ++i) c->fe[((int64_t)e * npe + a) * DM + i] = facc[a][i];
    }
}
void geom(femcy_ctx* c, const double* u, unsigned what) {
    const double t = c->opt_timing ? now_ms() : 0.0;
    if (c->dm == 3) geom_pass<3>(c, u, what); else geom_pass<2>(c, u, what);
    if (c->opt_timing) {
        c->timing.geom_ms += now_ms() - t;
        c->timing.geom_launches++;
    }
}

// assemble_nodal_force_GN_kernel (stiffnessMtrx.py:620-644): per node, its incident elements in ascending order
void nodal_force(femcy_ctx* c, double* f) {
    const double t = c->opt_timing ? now_ms() : 0.0;
    const int dm = c->dm;
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < c->nn; ++a) {
        double acc[3] = {0, 0, 0};
        for (int32_t k = c->ne_ptr[a]; k < c->ne_ptr[a + 1]; ++k)
            for (int i = 0; i < dm; ++i) acc[i] += c->fe[(int64_t)c->ne_idx[k] * dm + i];
        for (int i = 0; i < dm; ++i) f[(int64_t)a * dm + i] = acc[i];
    }
    if (c->opt_timing) {
        c->timing.force_ms += now_ms() - t;
        c->timing.force_launches++;
    }
}

// assemble_stiffnessMtrx (stiffnessMtrx.py:161-186): K_ab = sum_g B_a^T C B_b vol, owner-computes by row
template <int DM>
void assemble(femcy_ctx* c) {
    constexpr int DD = DM * DM;
    const int npe = c->npe, nGP = c->nGP;
    const bool consistent = c->opt_tangent == 1;
    const bool neo = c->mat_kind == FEMCY_MAT_NEOHOOKE;
    const int s = DM == 3 ? 6 : 3;
    const double lam = c->C[0 * s + 1], mu = c->C[(s - 1) * s + (s - 1)];
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t a = 0; a < c->nn; ++a) {
        double* row = c->K.data() + c->rowptr[a] * DD;
        std::fill(row, row + (c->rowptr[a + 1] - c->rowptr[a]) * DD, 0.0);
        for (int32_t k = c->ne_ptr[a]; k < c->ne_ptr[a + 1]; ++k) {
            const int32_t code = c->ne_idx[k];
            const int64_t e = code / npe;
            const int la = code - (int32_t)e * npe;
            for (int lb = 0; lb < npe; ++lb) {
                double blk[DD];
                for (int q = 0; q < DD; ++q) blk[q] = 0.0;
                for (int g = 0; g < nGP; ++g) {
                    const int64_t gp = e * nGP + g;
                    const double* ga = c->dsdx.data() + (gp * npe + la) * DM;
                    const double* gb = c->dsdx.data() + (gp * npe + lb) * DM;
                    if (consistent)
                        kblock_consistent<DM>(ga, gb, c->F.data() + gp * DD, c->sigma.data() + gp * DD, neo, lam, mu,
                                              c->params[0], c->params[1], c->vol[gp], blk);
                    else if (DM == 3 && c->C_is_cubic)
                        kblock_cubic3(ga, gb, c->cubic[0], c->cubic[1], c->cubic[2], c->vol[gp], reinterpret_cast<double(&)[9]>(blk));
                    else
                        kblock_add<DM>(ga, gb, c->C, c->vol[gp], blk);
                }
                double* dst = c->K.data() + c->eslot[((int64_t)e * npe + la) * npe + lb] * DD;
                for (int q = 0; q < DD; ++q) dst[q] += blk[q];
            }
        }
    }
}
int assemble_K(femcy_ctx* c) {
    if (c->opt_tangent == 1 && c->mat_kind == FEMCY_MAT_PSTRESS) {
        set_error("the consistent tangent is not available for plane stress");
        return FEMCY_EINVAL;
    }
    const double t = c->opt_timing ? now_ms() : 0.0;
    if (c->dm == 3) assemble<3>(c); else assemble<2>(c);
    if (c->opt_timing) {
        c->timing.assemble_ms += now_ms() - t;
        c->timing.assemble_launches++;
    }
    return FEMCY_OK;
}

// compute_Ad (conjugateGradientSolver.py:53-58); returns x.y when dot != nullptr (chunked: thread-count independent)
template <int DM>
void spmv_t(const femcy_ctx* c, const double* x, double* y, double* dot) {
    constexpr int DD = DM * DM;
    const int64_t nchunk = (c->nn + NODE_CHUNK - 1) / NODE_CHUNK;
    std::vector<double> part(dot ? (size_t)nchunk : 0, 0.0);
#pragma omp parallel for schedule(static)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
        double pd = 0.0;
        const int32_t a1 = (int32_t)std::min<int64_t>(c->nn, (ch + 1) * NODE_CHUNK);
        for (int32_t a = (int32_t)(ch * NODE_CHUNK); a < a1; ++a) {
            double acc[DM];
            for (int r = 0; r < DM; ++r) acc[r] = 0.0;
            for (int64_t p = c->rowptr[a]; p < c->rowptr[a + 1]; ++p) {
                const double* b = c->K.data() + p * DD;
                const double* xv = x + (int64_t)c->col[p] * DM;
                for (int r = 0; r < DM; ++r)
                    for (int cc = 0; cc < DM; ++cc) acc[r] += b[r * DM + cc] * xv[cc];
            }
            for (int r = 0; r < DM; ++r) {
                y[(int64_t)a * DM + r] = acc[r];
                pd += x[(int64_t)a * DM + r] * acc[r];
            }
        }
        if (dot) part[ch] = pd;
    }
    if (dot) {
        double sum = 0.0;
        for (double v : part) sum += v;
        *dot = sum;
    }
}
void spmv(femcy_ctx* c, const double* x, double* y, double* dot) {
    const double t = c->opt_timing ? now_ms() : 0.0;
    if (c->dm == 3) spmv_t<3>(c, x, y, dot); else spmv_t<2>(c, x, y, dot);
    if (c->opt_timing) {
        c->timing.spmv_ms += now_ms() - t;
        c->timing.spmv_launches++;
    }
}

double nan_to_inf_abs(double r) {
    const double a = std::fabs(r);
    return (a != a) ? INFINITY : a;
}

// dirichletBC_forNewtonMethod_kernel / the matrix part of dirichletBC_linearEquations (stiffnessMtrx.py:279-341): rows
// and columns of the constrained DOFs zeroed, unit diagonal, optionally the residual entry zeroed
void dirichlet_zero(femcy_ctx* c, const int32_t* dofs, int32_t k, double* resid) {
    const int dm = c->dm, dd = dm * dm;
    for (int32_t q = 0; q < k; ++q) {   // serial: different constrained DOFs touch the same blocks
        const int32_t dof = dofs[q], a = dof / dm, r = dof % dm;
        for (int64_t p = c->rowptr[a]; p < c->rowptr[a + 1]; ++p) {
            for (int cc = 0; cc < dm; ++cc) c->K[p * dd + r * dm + cc] = 0.0;
            const int32_t b = c->col[p];
            int64_t pm = -1;
            if (b == a) {
                pm = p;
            } else {
                const int32_t* lo = c->col.data() + c->rowptr[b] + 1;
                const int32_t* hi = c->col.data() + c->rowptr[b + 1];
                const int32_t* it = std::lower_bound(lo, hi, a);
                if (it != hi && *it == a) pm = it - c->col.data();
            }
            if (pm >= 0)
                for (int cc = 0; cc < dm; ++cc) c->K[pm * dd + cc * dm + r] = 0.0;
        }
        c->K[c->rowptr[a] * dd + r * dm + r] = 1.0;
        if (resid) resid[dof] = 0.0;
    }
}

int check_vec(femcy_ctx* c, int v) {
    if (v < 0 || v >= FEMCY_VEC_COUNT) {
        set_error("vector id %d out of range", v);
        return FEMCY_EINVAL;
    }
    if (c->vec[v].empty()) {
        set_error("vectors are allocated by femcy_set_mesh; call it first");
        return FEMCY_EINVAL;
    }
    return FEMCY_OK;
}

}  // namespace

#define CTX_OR_FAIL(ctx)              \
    if (!(ctx)) {                     \
        set_error("null context");    \
        return FEMCY_EINVAL;          \
    }                                 \
    femcy_ctx* c = (ctx)
#define VEC_OR_FAIL(v)               \
    {                                \
        int _rc = check_vec(c, (v)); \
        if (_rc) return _rc;         \
    }
#define READY_OR_FAIL()                                                                                         \
    REQUIRE(c->have_mesh&& c->have_element&& c->have_material&& c->have_pattern,                                \
            "context not fully defined (mesh=%d element=%d material=%d pattern=%d)", (int)c->have_mesh,        \
            (int)c->have_element, (int)c->have_material, (int)c->have_pattern)

extern "C" {

const char* femcy_last_error(void) { return g_err; }
int femcy_version(void) { return 100; }

int femcy_ctx_create(int device, femcy_ctx** out) {
    if (!out) {
        set_error("out is null");
        return FEMCY_EINVAL;
    }
    if (device != 0) {                              // one host = device 0 (the caller's bookkeeping stays honest)
        set_error("device %d out of range (the CPU backend has one device: 0)", device);
        return FEMCY_EINVAL;
    }
    *out = new (std::nothrow) femcy_ctx();
    if (!*out) return FEMCY_ENOMEM;
    return FEMCY_OK;
}
int femcy_ctx_destroy(femcy_ctx* ctx) {
    delete ctx;
    return FEMCY_OK;
}

int femcy_set_option(femcy_ctx* ctx, int option, int64_t value) {
    CTX_OR_FAIL(ctx);
    switch (option) {
        case FEMCY_OPT_TANGENT:
            REQUIRE(value == 0 || value == 1, "tangent: 0 (reference) or 1 (consistent)");
            c->opt_tangent = (int)value;
            return FEMCY_OK;
        case FEMCY_OPT_TIMING:
            c->opt_timing = value > 0 ? 1 : 0;
            return FEMCY_OK;
        case FEMCY_OPT_SELL_SIGMA:
            REQUIRE(!c->have_pattern, "set the sorting window before femcy_build_pattern");
            return FEMCY_OK;
        // schedule / layout knobs of the device kernels: accepted, without effect on the host
        case FEMCY_OPT_ASSEMBLY: case FEMCY_OPT_PCG_POLL: case FEMCY_OPT_SPMV_VARIANT: case FEMCY_OPT_EW_GRID:
        case FEMCY_OPT_PCG_GRAPH: case FEMCY_OPT_PCG_PERSIST: case FEMCY_OPT_PCG_SMALL: case FEMCY_OPT_OVERLAP:
        case FEMCY_OPT_PCG_PERSIST_MULTI: case FEMCY_OPT_PCG_STORAGE_ORDER: case FEMCY_OPT_PCG_FUSED_UPDATE: case FEMCY_OPT_SPMV_FOOTPRINT:
            return FEMCY_OK;
        case FEMCY_OPT_NODE_ORDER:
            REQUIRE(!c->have_pattern, "set the node order before femcy_build_pattern");
            return FEMCY_OK;
        case FEMCY_OPT_DIRECT_MAX_BYTES:
            REQUIRE(value >= (int64_t)1 << 20, "direct solve: the band limit must be at least 1 MiB");
            c->direct_max_bytes = value;
            return FEMCY_OK;
        default:
            if (option >= 100 && option <= 119) return FEMCY_OK;   // FEMCY_TUNE_*: device tuning knobs
            set_error("unknown option %d", option);
            return FEMCY_EINVAL;
    }
}
int femcy_sync(femcy_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    (void)c;
    return FEMCY_OK;
}

int femcy_set_mesh(femcy_ctx* ctx, int32_t nn, int32_t dm, const double* nodes, int32_t ne, int32_t npe,
                   const int32_t* elems) {
    CTX_OR_FAIL(ctx);
    REQUIRE(nodes && elems, "null mesh arrays");
    REQUIRE(nn > 0 && ne > 0, "empty mesh (nn=%d, ne=%d)", nn, ne);
    REQUIRE(dm == 2 || dm == 3, "dm must be 2 or 3, got %d", dm);
    REQUIRE(npe >= 2 && npe <= 27, "npe out of range: %d", npe);
    for (int64_t k = 0; k < (int64_t)ne * npe; ++k)
        REQUIRE(elems[k] >= 0 && elems[k] < nn, "element %lld references node %d outside [0,%d)", (long long)(k / npe),
                elems[k], nn);
    c->dofsets.clear();
    c->loadsets.clear();
    c->nn = nn; c->dm = dm; c->ne = ne; c->npe = npe;
    c->n = (int64_t)nn * dm;
    c->nodes.assign(nodes, nodes + (size_t)nn * dm);
    c->elems.assign(elems, elems + (size_t)ne * npe);
    for (auto& v : c->vec) v.assign((size_t)c->n, 0.0);
    c->r.assign((size_t)c->n, 0.0);
    c->d.assign((size_t)c->n, 0.0);
    c->M.assign((size_t)c->n, 0.0);
    c->Ad.assign((size_t)c->n, 0.0);
    c->have_mesh = true;
    c->have_material = c->have_element = c->have_pattern = false;
    return FEMCY_OK;
}

int femcy_set_element(femcy_ctx* ctx, int32_t nGP, const double* dN, const double* w, int32_t voigt_kind) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh, "femcy_set_mesh must come first");
    REQUIRE(dN && w && nGP >= 1 && nGP <= 64, "bad element tables (nGP=%d)", nGP);
    REQUIRE((voigt_kind == FEMCY_VOIGT_2D && c->dm == 2) || (voigt_kind == FEMCY_VOIGT_3D && c->dm == 3),
            "voigt kind %d does not match dm=%d", voigt_kind, c->dm);
    c->nGP = nGP;
    c->s = c->dm == 2 ? 3 : 6;
    c->dN.assign(dN, dN + (size_t)nGP * c->npe * c->dm);
    c->w.assign(w, w + nGP);
    const size_t ngp = (size_t)c->ne * nGP, dd = (size_t)c->dm * c->dm;
    c->dsdx.assign(ngp * c->npe * c->dm, 0.0);
    c->vol.assign(ngp, 0.0);
    c->F.assign(ngp * dd, 0.0);
    c->sigma.assign(ngp * dd, 0.0);
    c->strain.assign(ngp * dd, 0.0);
    c->mises.assign(ngp, 0.0);
    c->energy.assign(ngp, 0.0);
    c->fe.assign((size_t)c->ne * c->npe * c->dm, 0.0);
    c->have_element = true;
    return FEMCY_OK;
}

int femcy_set_material(femcy_ctx* ctx, int32_t kind, const double* C, const double* params, int32_t nparams) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh, "femcy_set_mesh must come first");
    REQUIRE(C, "null C");
    REQUIRE(kind >= FEMCY_MAT_LIN3D && kind <= FEMCY_MAT_NEOHOOKE, "unknown material kind %d", kind);
    const bool ok_dm = kind == FEMCY_MAT_NEOHOOKE || ((kind == FEMCY_MAT_LIN3D) == (c->dm == 3));
    REQUIRE(ok_dm, "material kind %d does not match dm=%d", kind, c->dm);
    REQUIRE(nparams >= 2 && params, "material needs 2 parameters");
    const int s = c->dm == 2 ? 3 : 6;
    for (int i = 0; i < s * s; ++i) c->C[i] = C[i];
    c->mat_kind = kind;
    c->C_is_cubic = false;
    if (s == 6) {   // cubic pattern (exact comparisons), as on the device
        const double c11 = C[0], c12 = C[1], c44 = C[3 * 6 + 3];
        bool ok = true;
        for (int i = 0; i < 6 && ok; ++i)
            for (int j = 0; j < 6 && ok; ++j) {
                const double want = (i == j) ? (i < 3 ? c11 : c44) : ((i < 3 && j < 3) ? c12 : 0.0);
                ok = C[i * 6 + j] == want;
            }
        c->C_is_cubic = ok;
        c->cubic[0] = c11; c->cubic[1] = c12; c->cubic[2] = c44;
    }
    for (int i = 0; i < 4; ++i) c->params[i] = (i < nparams) ? params[i] : 0.0;
    c->have_material = true;
    return FEMCY_OK;
}

// body.get_nodeEles / get_coElement_nodes + sparseIJ (body.py:165-194, stiffnessMtrx.py:70-89)
int femcy_build_pattern(femcy_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh, "femcy_set_mesh must come first");
    const int32_t nn = c->nn, ne = c->ne, npe = c->npe;
    c->ne_ptr.assign((size_t)nn + 1, 0);
    for (int64_t k = 0; k < (int64_t)ne * npe; ++k) c->ne_ptr[c->elems[k] + 1]++;
    c->max_node_elems = 0;
    for (int32_t a = 0; a < nn; ++a) {
        c->max_node_elems = std::max(c->max_node_elems, c->ne_ptr[a + 1]);
        c->ne_ptr[a + 1] += c->ne_ptr[a];
    }
    c->ne_idx.resize((size_t)ne * npe);
    {
        std::vector<int32_t> cur(c->ne_ptr.begin(), c->ne_ptr.end() - 1);
        for (int64_t k = 0; k < (int64_t)ne * npe; ++k) c->ne_idx[cur[c->elems[k]]++] = (int32_t)k;   // ascending (e, la)
    }
    // node adjacency: diagonal first, then ascending
    std::vector<int32_t> rowlen((size_t)nn);
    std::vector<std::vector<int32_t>> rows((size_t)nn);
#pragma omp parallel
    {
        std::vector<int32_t> tmp;
#pragma omp for schedule(dynamic, 256)
        for (int32_t a = 0; a < nn; ++a) {
            tmp.clear();
            for (int32_t k = c->ne_ptr[a]; k < c->ne_ptr[a + 1]; ++k) {
                const int64_t e = c->ne_idx[k] / npe;
                for (int lb = 0; lb < npe; ++lb) tmp.push_back(c->elems[e * npe + lb]);
            }
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            auto& row = rows[a];
            row.clear();
            row.push_back(a);
            for (int32_t b : tmp)
                if (b != a) row.push_back(b);
            rowlen[a] = (int32_t)row.size();
        }
    }
    c->rowptr.assign((size_t)nn + 1, 0);
    c->max_row_blocks = 0;
    for (int32_t a = 0; a < nn; ++a) {
        c->rowptr[a + 1] = c->rowptr[a] + rowlen[a];
        c->max_row_blocks = std::max(c->max_row_blocks, rowlen[a]);
    }
    const int64_t nnzb = c->rowptr[nn];
    c->col.resize((size_t)nnzb);
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < nn; ++a) std::copy(rows[a].begin(), rows[a].end(), c->col.begin() + c->rowptr[a]);
    rows.clear();
    rows.shrink_to_fit();
    c->K.assign((size_t)nnzb * c->dm * c->dm, 0.0);
    c->eslot.resize((size_t)ne * npe * npe);
#pragma omp parallel for schedule(static)
    for (int32_t e = 0; e < ne; ++e)
        for (int la = 0; la < npe; ++la) {
            const int32_t a = c->elems[(int64_t)e * npe + la];
            const int32_t* lo = c->col.data() + c->rowptr[a];
            const int32_t* hi = c->col.data() + c->rowptr[a + 1];
            for (int lb = 0; lb < npe; ++lb) {
                const int32_t b = c->elems[(int64_t)e * npe + lb];
                const int64_t p = (b == a) ? c->rowptr[a] : (std::lower_bound(lo + 1, hi, b) - c->col.data());
                c->eslot[((int64_t)e * npe + la) * npe + lb] = p;
            }
        }
    c->have_pattern = true;
    c->have_band_order = false;
    return FEMCY_OK;
}

int femcy_get_pattern_info(femcy_ctx* ctx, femcy_pattern_info* out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern && out, "pattern not built");
    out->n = c->n;
    out->nnzb = c->rowptr[c->nn];
    out->nnz = out->nnzb * c->dm * c->dm;
    out->max_row_blocks = c->max_row_blocks;
    out->ell_width = c->max_row_blocks * c->dm;
    out->stored_blocks = out->nnzb;                 // block-CSR: no padding
    out->nslices = (c->nn + 63) / 64;
    out->max_node_elems = c->max_node_elems;
    return FEMCY_OK;
}
int femcy_get_node_order(femcy_ctx* ctx, int32_t* used, double* lines) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern, "pattern not built");
    if (used) *used = 0;                                  // the host backend keeps the caller's numbering (block CSR)
    if (lines)
        for (int k = 0; k < 7; ++k) lines[k] = 0.0;
    return FEMCY_OK;
}

// ---------------------------------------------------------------------------- vector plumbing
int femcy_vec_upload(femcy_ctx* ctx, int vec, const double* src, int64_t n) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    REQUIRE(src && n == c->n, "upload length %lld != n = %lld", (long long)n, (long long)c->n);
    std::memcpy(c->vec[vec].data(), src, sizeof(double) * n);
    return FEMCY_OK;
}
int femcy_vec_download(femcy_ctx* ctx, int vec, double* dst, int64_t n) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    REQUIRE(dst && n == c->n, "download length %lld != n = %lld", (long long)n, (long long)c->n);
    std::memcpy(dst, c->vec[vec].data(), sizeof(double) * n);
    return FEMCY_OK;
}
int femcy_vec_fill(femcy_ctx* ctx, int vec, double value) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    std::fill(c->vec[vec].begin(), c->vec[vec].end(), value);
    return FEMCY_OK;
}
int femcy_vec_copy(femcy_ctx* ctx, int dst, int src) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(dst);
    VEC_OR_FAIL(src);
    if (dst != src) c->vec[dst] = c->vec[src];
    return FEMCY_OK;
}
int femcy_vec_scatter(femcy_ctx* ctx, int vec, const int32_t* idx, const double* vals, int32_t k) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    if (k == 0) return FEMCY_OK;
    REQUIRE(idx && vals && k > 0, "bad scatter arguments");
    for (int32_t i = 0; i < k; ++i) REQUIRE(idx[i] >= 0 && idx[i] < c->n, "scatter index %d out of range", idx[i]);
    for (int32_t i = 0; i < k; ++i) c->vec[vec][idx[i]] = vals[i];
    return FEMCY_OK;
}
int femcy_vec_sub(femcy_ctx* ctx, int cv, int a, int b) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(cv); VEC_OR_FAIL(a); VEC_OR_FAIL(b);
    double* pc = c->vec[cv].data();
    const double *pa = c->vec[a].data(), *pb = c->vec[b].data();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < c->n; ++i) pc[i] = pa[i] - pb[i];
    return FEMCY_OK;
}
int femcy_vec_axpy(femcy_ctx* ctx, int a, int b, double cc, int d) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(a); VEC_OR_FAIL(b); VEC_OR_FAIL(d);
    double* pa = c->vec[a].data();
    const double *pb = c->vec[b].data(), *pd = c->vec[d].data();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < c->n; ++i) pa[i] = pb[i] + cc * pd[i];
    return FEMCY_OK;
}
int femcy_vec_scale(femcy_ctx* ctx, int vec, double s) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    for (double& v : c->vec[vec]) v *= s;
    return FEMCY_OK;
}
int femcy_vec_norm(femcy_ctx* ctx, int vec, double* rms) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    REQUIRE(rms, "null output");
    const double* p = c->vec[vec].data();
    const int64_t nchunk = (c->n + CHUNK - 1) / CHUNK;
    std::vector<double> part((size_t)nchunk);
#pragma omp parallel for schedule(static)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
        double s = 0.0;
        const int64_t i1 = std::min(c->n, (ch + 1) * CHUNK);
        for (int64_t i = ch * CHUNK; i < i1; ++i) s += p[i] * p[i];
        part[ch] = s;
    }
    double ss = 0.0;
    for (double v : part) ss += v;
    *rms = std::sqrt(ss / (double)c->n);
    return FEMCY_OK;
}
int femcy_vec_absmax(femcy_ctx* ctx, int vec, double* out) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    REQUIRE(out, "null output");
    double m = 0.0;
    for (double v : c->vec[vec]) m = std::fmax(m, nan_to_inf_abs(v));
    *out = m;
    return FEMCY_OK;
}

// -------------------------------------------------------------------------------- the hot path
int femcy_assemble_K(femcy_ctx* ctx, int u_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    const double* u = nullptr;
    if (u_vec >= 0) {
        VEC_OR_FAIL(u_vec);
        u = c->vec[u_vec].data();
    }
    geom(c, u, c->opt_tangent == 1 ? (GEOM_DSDX | GEOM_F | GEOM_SIGMA) : GEOM_DSDX);
    return assemble_K(c);
}
int femcy_internal_force(femcy_ctx* ctx, int u_vec, int f_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(u_vec);
    VEC_OR_FAIL(f_vec);
    geom(c, c->vec[u_vec].data(), GEOM_DSDX | GEOM_F | GEOM_SIGMA | GEOM_FE);
    nodal_force(c, c->vec[f_vec].data());
    return FEMCY_OK;
}
int femcy_residual_and_K(femcy_ctx* ctx, int u_vec, int f_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(u_vec);
    VEC_OR_FAIL(f_vec);
    geom(c, c->vec[u_vec].data(), GEOM_DSDX | GEOM_F | GEOM_SIGMA | GEOM_FE);
    nodal_force(c, c->vec[f_vec].data());
    return assemble_K(c);
}

static int check_dofs(femcy_ctx* c, const int32_t* dofs, int32_t k) {
    for (int32_t i = 0; i < k; ++i)
        if (dofs[i] < 0 || dofs[i] >= c->n) {
            set_error("constrained DOF %d out of range", dofs[i]);
            return FEMCY_EINVAL;
        }
    return FEMCY_OK;
}
// dirichletBC_linearEquations (stiffnessMtrx.py:279-307), race-free: rhs -= K s with the not yet modified matrix
static int dirichlet_linear(femcy_ctx* c, const int32_t* dofs, const double* vals, double value, int32_t k, int rhs_vec) {
    bool any = false;
    for (int32_t i = 0; i < k; ++i) any = any || ((vals ? vals[i] : value) != 0.0);
    double* rhs = c->vec[rhs_vec].data();
    if (any) {
        std::vector<double>& s = c->vec[FEMCY_VEC_TMP0];
        std::vector<double>& Ks = c->vec[FEMCY_VEC_TMP1];
        std::fill(s.begin(), s.end(), 0.0);
        for (int32_t i = 0; i < k; ++i) s[dofs[i]] = vals ? vals[i] : value;
        spmv(c, s.data(), Ks.data(), nullptr);
        for (int64_t i = 0; i < c->n; ++i) rhs[i] -= Ks[i];
    }
    for (int32_t i = 0; i < k; ++i) rhs[dofs[i]] = vals ? vals[i] : value;
    dirichlet_zero(c, dofs, k, nullptr);
    return FEMCY_OK;
}
int femcy_apply_dirichlet_linear(femcy_ctx* ctx, const int32_t* dofs, const double* vals, int32_t k, int rhs_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(rhs_vec);
    if (k == 0) return FEMCY_OK;
    REQUIRE(k > 0 && dofs && vals, "bad Dirichlet arguments");
    REQUIRE(rhs_vec != FEMCY_VEC_TMP0 && rhs_vec != FEMCY_VEC_TMP1, "rhs may not alias the scratch vectors");
    int rc = check_dofs(c, dofs, k);
    if (rc) return rc;
    return dirichlet_linear(c, dofs, vals, 0.0, k, rhs_vec);
}
int femcy_apply_dirichlet_newton(femcy_ctx* ctx, const int32_t* dofs, int32_t k, int residual_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(residual_vec);
    if (k == 0) return FEMCY_OK;
    REQUIRE(dofs && k > 0, "bad Dirichlet arguments");
    int rc = check_dofs(c, dofs, k);
    if (rc) return rc;
    dirichlet_zero(c, dofs, k, c->vec[residual_vec].data());
    return FEMCY_OK;
}
int femcy_dofset_create(femcy_ctx* ctx, const int32_t* dofs, int32_t k, int32_t* id_out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh && id_out && k >= 0 && (k == 0 || dofs), "bad dofset arguments");
    for (int32_t i = 0; i < k; ++i) REQUIRE(dofs[i] >= 0 && dofs[i] < c->n, "DOF %d out of range", dofs[i]);
    DofSet ds;
    if (k) ds.dofs.assign(dofs, dofs + k);
    c->dofsets.push_back(ds);
    *id_out = (int32_t)c->dofsets.size() - 1;
    return FEMCY_OK;
}
#define DOFSET_OR_FAIL(id)                                                                      \
    REQUIRE((id) >= 0 && (size_t)(id) < c->dofsets.size(), "unknown dofset %d", (int)(id)); \
    const DofSet& ds = c->dofsets[(id)]
int femcy_dofset_dirichlet_newton(femcy_ctx* ctx, int32_t id, int residual_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(residual_vec);
    DOFSET_OR_FAIL(id);
    dirichlet_zero(c, ds.dofs.data(), (int32_t)ds.dofs.size(), c->vec[residual_vec].data());
    return FEMCY_OK;
}
int femcy_dofset_dirichlet_linear(femcy_ctx* ctx, int32_t id, double value, int rhs_vec) {
    CTX_OR_FAIL(ctx);
    READY_OR_FAIL();
    VEC_OR_FAIL(rhs_vec);
    DOFSET_OR_FAIL(id);
    REQUIRE(rhs_vec != FEMCY_VEC_TMP0 && rhs_vec != FEMCY_VEC_TMP1, "rhs may not alias the scratch vectors");
    if (ds.dofs.empty()) return FEMCY_OK;
    return dirichlet_linear(c, ds.dofs.data(), nullptr, value, (int32_t)ds.dofs.size(), rhs_vec);
}
int femcy_dofset_fill(femcy_ctx* ctx, int32_t id, int vec, double value) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    DOFSET_OR_FAIL(id);
    for (int32_t dof : ds.dofs) c->vec[vec][dof] = value;
    return FEMCY_OK;
}
int femcy_dofset_scatter(femcy_ctx* ctx, int32_t id, int vec, const double* vals) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(vec);
    DOFSET_OR_FAIL(id);
    if (ds.dofs.empty()) return FEMCY_OK;
    REQUIRE(vals, "null values");
    for (size_t i = 0; i < ds.dofs.size(); ++i) c->vec[vec][ds.dofs[i]] = vals[i];
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------- Neumann load sets
int femcy_loadset_create(femcy_ctx* ctx, int32_t nft, int32_t nfn, int32_t nip, const int32_t* ft_nodes,
                         const double* ft_N, const double* ft_dN, const double* ft_normal, const double* ft_weight,
                         int32_t nload, const int32_t* load_elem, const int32_t* load_ft, int32_t* id_out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh && id_out, "mesh not set or null id_out");
    REQUIRE(nft > 0 && nip > 0 && nfn >= c->dm && nfn <= c->npe, "bad facet table sizes (nft %d, nfn %d, nip %d)", nft, nfn, nip);
    REQUIRE(ft_nodes && ft_N && ft_dN && ft_normal && ft_weight, "null facet tables");
    REQUIRE(nload >= 0 && (nload == 0 || (load_elem && load_ft)), "bad load facet lists");
    for (int32_t i = 0; i < nft * nfn; ++i)
        REQUIRE(ft_nodes[i] >= 0 && ft_nodes[i] < c->npe, "facet table: local node %d out of range", ft_nodes[i]);
    for (int32_t l = 0; l < nload; ++l) {
        REQUIRE(load_elem[l] >= 0 && load_elem[l] < c->ne, "load facet %d: element %d out of range", l, load_elem[l]);
        REQUIRE(load_ft[l] >= 0 && load_ft[l] < nft, "load facet %d: facet type %d out of range", l, load_ft[l]);
    }
    LoadSet ls;
    ls.nft = nft; ls.nfn = nfn; ls.nip = nip; ls.nload = nload;
    const size_t tip = (size_t)nft * nip, nslot = (size_t)nload * nfn;
    ls.ft_nodes.assign(ft_nodes, ft_nodes + (size_t)nft * nfn);
    ls.N.assign(ft_N, ft_N + tip * c->npe);
    ls.dN.assign(ft_dN, ft_dN + tip * c->npe * c->dm);
    ls.normal.assign(ft_normal, ft_normal + tip * c->dm);
    ls.weight.assign(ft_weight, ft_weight + tip);
    if (nload) {
        ls.elem.assign(load_elem, load_elem + nload);
        ls.ft.assign(load_ft, load_ft + nload);
    }
    // loaded nodes and, per node, its contribution slots (facet * nfn + facet node) in ascending order
    std::vector<int32_t> slot_node(nslot);
    ls.slot.resize(nslot);
    for (int32_t l = 0; l < nload; ++l)
        for (int32_t f = 0; f < nfn; ++f)
            slot_node[(size_t)l * nfn + f] = c->elems[(size_t)load_elem[l] * c->npe + ft_nodes[(size_t)load_ft[l] * nfn + f]];
    for (size_t i = 0; i < nslot; ++i) ls.slot[i] = (int32_t)i;
    std::stable_sort(ls.slot.begin(), ls.slot.end(), [&](int32_t a, int32_t b) { return slot_node[a] < slot_node[b]; });
    for (size_t i = 0; i < nslot; ++i)
        if (i == 0 || slot_node[ls.slot[i]] != slot_node[ls.slot[i - 1]]) {
            ls.node.push_back(slot_node[ls.slot[i]]);
            ls.ptr.push_back((int32_t)i);
        }
    ls.ptr.push_back((int32_t)nslot);
    ls.contrib.assign(std::max<size_t>(nslot, 1) * c->dm, 0.0);
    c->loadsets.push_back(std::move(ls));
    *id_out = (int32_t)c->loadsets.size() - 1;
    return FEMCY_OK;
}

// neumannBC (stiffnessMtrx.py:369-411)
int femcy_loadset_neumann(femcy_ctx* ctx, int32_t id, double traction, const double* direction, int rhs_vec) {
    CTX_OR_FAIL(ctx);
    VEC_OR_FAIL(rhs_vec);
    REQUIRE(id >= 0 && (size_t)id < c->loadsets.size(), "unknown load set %d", (int)id);
    LoadSet& ls = c->loadsets[id];
    double* rhs = c->vec[rhs_vec].data();
    std::fill(rhs, rhs + c->n, 0.0);                // reference :384
    const int dm = c->dm;
#pragma omp parallel for schedule(static)
    for (int32_t l = 0; l < ls.nload; ++l) {
        const int32_t* en = c->elems.data() + (int64_t)ls.elem[l] * c->npe;
        double* out = ls.contrib.data() + (int64_t)l * ls.nfn * dm;
        if (dm == 3)
            neumann_facet<3>(c->npe, ls.nfn, ls.nip, c->nodes.data(), en, ls.ft[l], ls.ft_nodes.data(), ls.N.data(),
                             ls.dN.data(), ls.normal.data(), ls.weight.data(), traction, direction, out);
        else
            neumann_facet<2>(c->npe, ls.nfn, ls.nip, c->nodes.data(), en, ls.ft[l], ls.ft_nodes.data(), ls.N.data(),
                             ls.dN.data(), ls.normal.data(), ls.weight.data(), traction, direction, out);
    }
    for (size_t i = 0; i < ls.node.size(); ++i)
        for (int dd = 0; dd < dm; ++dd) {
            double s = 0.0;
            for (int32_t k = ls.ptr[i]; k < ls.ptr[i + 1]; ++k) s += ls.contrib[(int64_t)ls.slot[k] * dm + dd];
            rhs[(int64_t)ls.node[i] * dm + dd] = s;
        }
    return FEMCY_OK;
}

int femcy_spmv(femcy_ctx* ctx, int x_vec, int y_vec) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern, "pattern not built");
    VEC_OR_FAIL(x_vec);
    VEC_OR_FAIL(y_vec);
    REQUIRE(x_vec != y_vec, "spmv cannot run in place");
    spmv(c, c->vec[x_vec].data(), c->vec[y_vec].data(), nullptr);
    return FEMCY_OK;
}

int femcy_direct_plan(femcy_ctx* ctx, femcy_direct_info* info) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern, "pattern not built");
    REQUIRE(info != nullptr, "femcy_direct_plan: info must not be null");
    if (!c->have_band_order) {
        c->band_order = band_order_rcm(c->nn, c->ne, c->npe, c->elems.data());
        c->have_band_order = true;
    }
    *info = femcy_direct_info{};
    const int64_t bw = ((int64_t)c->band_order.half_band_nodes + 1) * c->dm - 1;
    info->n = c->n;
    info->bandwidth = (int32_t)bw;
    info->band_bytes = (int64_t)((double)c->n * (double)(bw + 1) * 8.0);
    return FEMCY_OK;
}

// solve_by_scipy (stiffnessMtrx.py:219-251): direct solve -- reverse Cuthill-McKee, lower band by columns, K = L S L^T,
// residual check + refinement (csrc/band_order.hpp; the device library does the same on tiles, csrc/kernels_direct.hip)
int femcy_direct_solve(femcy_ctx* ctx, int b_vec, int x_vec, femcy_direct_info* info) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern, "pattern not built");
    VEC_OR_FAIL(b_vec);
    VEC_OR_FAIL(x_vec);
    REQUIRE(b_vec != x_vec, "direct solve: b and x must be different vectors");
    if (!c->have_band_order) {
        c->band_order = band_order_rcm(c->nn, c->ne, c->npe, c->elems.data());
        c->have_band_order = true;
    }
    const BandOrder& o = c->band_order;
    const int dm = c->dm, dd = dm * dm;
    const int64_t n = c->n, bw = ((int64_t)o.half_band_nodes + 1) * dm - 1, w = bw + 1;
    const double bytes = (double)n * (double)w * 8.0;
    femcy_direct_info local{};
    if (!info) info = &local;
    *info = femcy_direct_info{};
    info->n = n;
    info->band_bytes = (int64_t)bytes;
    info->bandwidth = (int32_t)bw;
    if (bytes > (double)c->direct_max_bytes) {
        set_error("direct solve: the band of this system (%lld DOF, %lld sub-diagonals after reverse Cuthill-McKee) takes "
                  "%.1f GB, more than the limit of %.1f GB (FEMCY_OPT_DIRECT_MAX_BYTES)",
                  (long long)n, (long long)bw, bytes * 1e-9, (double)c->direct_max_bytes * 1e-9);
        return FEMCY_ENOMEM;
    }
    std::vector<double> A, sgn((size_t)n);
    try {
        A.assign((size_t)n * (size_t)w, 0.0);
    } catch (const std::bad_alloc&) {
        set_error("direct solve: out of memory (%.1f GB)", bytes * 1e-9);
        return FEMCY_ENOMEM;
    }
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < c->nn; ++a)
        for (int64_t q = c->rowptr[a]; q < c->rowptr[a + 1]; ++q) {
            const int32_t b = c->col[q];
            const int64_t ra = o.rank[a], rb = o.rank[b];
            if (ra < rb) continue;
            for (int r = 0; r < dm; ++r)
                for (int cc = 0; cc < dm; ++cc) {
                    const int64_t i = ra * dm + r, j = rb * dm + cc;
                    if (i >= j) A[(size_t)(j * w + (i - j))] = c->K[q * dd + r * dm + cc];
                }
        }
    int64_t negative = 0;
    const int64_t bad = band_factor_host(n, bw, A.data(), sgn.data(), &negative);
    info->negative_pivots = (int32_t)std::min<int64_t>(negative, INT32_MAX);
    if (bad) {
        info->singular_at = (int32_t)bad;
        set_error("direct solve: pivot %lld (band order) is zero or not a number -- the matrix is singular",
                  (long long)(bad - 1));
        return FEMCY_ENUMERIC;
    }
    const double* bvec = c->vec[b_vec].data();
    double* x = c->vec[x_vec].data();
    std::vector<double> y((size_t)n), res((size_t)n), Kx((size_t)n);
    auto solve_into = [&](const double* rhs, double* out, bool add) {      // out (+)= K^-1 rhs
        for (int64_t i = 0; i < n; ++i) y[(size_t)i] = rhs[(int64_t)o.node_at[i / dm] * dm + i % dm];
        band_solve_host(n, bw, A.data(), sgn.data(), y.data());
        for (int64_t i = 0; i < n; ++i) {
            double& dst = out[(int64_t)o.node_at[i / dm] * dm + i % dm];
            dst = add ? dst + y[(size_t)i] : y[(size_t)i];
        }
    };
    auto residual = [&]() {                                                 // res = b - K x; -> max|res| / max|b|
        spmv(c, x, Kx.data(), nullptr);
        double rm = 0.0, bm = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            res[(size_t)i] = bvec[i] - Kx[(size_t)i];
            rm = std::fmax(rm, std::fabs(res[(size_t)i]));
            bm = std::fmax(bm, std::fabs(bvec[i]));
        }
        if (std::isnan(rm)) return rm;
        return bm > 0.0 ? rm / bm : rm;
    };
    solve_into(bvec, x, false);
    double rel = residual();
    while (info->refinements < DIRECT_MAX_REFINE && rel > DIRECT_REFINE_ABOVE) {
        solve_into(res.data(), x, true);
        ++info->refinements;
        const double rel2 = residual();
        const bool stalled = !(rel2 < 0.5 * rel);
        rel = rel2;
        if (stalled) break;
    }
    info->residual = rel;
    if (!(rel <= DIRECT_ACCEPT)) {
        set_error("direct solve: residual %.3e max|b| after %d refinement steps (%d negative pivots): elimination without "
                  "pivoting lost this matrix", rel, info->refinements, info->negative_pivots);
        return FEMCY_ENUMERIC;
    }
    return FEMCY_OK;
}

// ConjugateGradientSolver_rowMajor.solve (conjugateGradientSolver.py:103-127)
int femcy_pcg(femcy_ctx* ctx, int b_vec, int x_vec, double eps, int32_t maxit, int32_t* iters, double* rmax0,
              double* rmax_out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern, "pattern not built");
    VEC_OR_FAIL(b_vec);
    VEC_OR_FAIL(x_vec);
    REQUIRE(b_vec != x_vec, "pcg: b and x must be different vectors");
    if (maxit <= 0) maxit = (int32_t)std::min<int64_t>(c->n, INT32_MAX);
    const double t_start = now_ms();
    const int64_t n = c->n;
    const int dm = c->dm, dd = dm * dm;
    const double* b = c->vec[b_vec].data();
    double *x = c->vec[x_vec].data(), *r = c->r.data(), *d = c->d.data(), *M = c->M.data(), *Ad = c->Ad.data();
    const int64_t nchunk = (n + CHUNK - 1) / CHUNK;
    std::vector<double> ps((size_t)nchunk), pm((size_t)nchunk);
    // M_init (:48-51), re_init + r_d_init (:32-38, 60-65)
#pragma omp parallel for schedule(static)
    for (int32_t a = 0; a < c->nn; ++a)
        for (int q = 0; q < dm; ++q) M[(int64_t)a * dm + q] = 1.0 / c->K[c->rowptr[a] * dd + q * dm + q];
    auto reduce_pair = [&](double& s, double& m) {
        s = 0.0;
        m = 0.0;
        for (int64_t ch = 0; ch < nchunk; ++ch) {
            s += ps[ch];
            m = std::fmax(m, pm[ch]);
        }
    };
#pragma omp parallel for schedule(static)
    for (int64_t ch = 0; ch < nchunk; ++ch) {
        double s = 0.0, m = 0.0;
        const int64_t i1 = std::min(n, (ch + 1) * CHUNK);
        for (int64_t i = ch * CHUNK; i < i1; ++i) {
            x[i] = 0.0;
            r[i] = b[i];
            d[i] = M[i] * b[i];
            s += b[i] * M[i] * b[i];
            m = std::fmax(m, nan_to_inf_abs(b[i]));
        }
        ps[ch] = s;
        pm[ch] = m;
    }
    double rMr, r0;
    reduce_pair(rMr, r0);
    double rmax = r0;
    int done = (r0 == 0.0) ? 1 : ((r0 != r0 || std::isinf(r0)) ? 2 : 0);
    int32_t it = 0;
    while (!done && it < maxit) {
        double dAd = 0.0;
        spmv(c, d, Ad, &dAd);
        const double alpha = rMr / dAd;
#pragma omp parallel for schedule(static)
        for (int64_t ch = 0; ch < nchunk; ++ch) {
            double s = 0.0, m = 0.0;
            const int64_t i1 = std::min(n, (ch + 1) * CHUNK);
            for (int64_t i = ch * CHUNK; i < i1; ++i) {
                x[i] += alpha * d[i];
                const double ri = r[i] - alpha * Ad[i];
                r[i] = ri;
                s += ri * M[i] * ri;
                m = std::fmax(m, nan_to_inf_abs(ri));
            }
            ps[ch] = s;
            pm[ch] = m;
        }
        double rMr_new;
        reduce_pair(rMr_new, rmax);
        ++it;
        if (rmax != rmax || std::isinf(rmax) || rMr_new != rMr_new) {
            done = 2;
        } else if (rmax < eps * r0) {
            done = 1;
        } else {
            const double beta = rMr_new / rMr;
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) d[i] = M[i] * r[i] + beta * d[i];
        }
        rMr = rMr_new;
    }
    if (iters) *iters = it;
    if (rmax0) *rmax0 = r0;
    if (rmax_out) *rmax_out = rmax;
    c->timing.pcg_iters += it;
    c->timing.solves_three++;
    if (c->opt_timing) c->timing.pcg_ms += now_ms() - t_start;
    if (done == 2) {
        set_error("PCG breakdown: NaN/Inf residual after %d iterations (r0 = %g)", it, r0);
        return FEMCY_ENUMERIC;
    }
    return FEMCY_OK;
}

// ------------------------------------------------------------------------------ post-processing
int femcy_compute_strain_stress(femcy_ctx* ctx, int u_vec, int large) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh && c->have_element && c->have_material, "context not fully defined");
    VEC_OR_FAIL(u_vec);
    geom(c, c->vec[u_vec].data(), GEOM_F);   // F only: dsdx, vol and (nlgeom) the stress of the last force evaluation stay
    const int64_t ngp = (int64_t)c->ne * c->nGP;
    const int dd = c->dm * c->dm;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < ngp; ++t) {
        if (c->dm == 3)
            post_point<3>(large ? 1 : 0, c->mat_kind, c->C, c->params[0], c->params[1], c->F.data() + t * dd,
                          c->sigma.data() + t * dd, c->strain.data() + t * dd, c->mises.data() + t);
        else
            post_point<2>(large ? 1 : 0, c->mat_kind, c->C, c->params[0], c->params[1], c->F.data() + t * dd,
                          c->sigma.data() + t * dd, c->strain.data() + t * dd, c->mises.data() + t);
    }
    return FEMCY_OK;
}

int femcy_elastic_energy(femcy_ctx* ctx, int u_vec, double* total) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_mesh && c->have_element && c->have_material && total, "context not fully defined");
    VEC_OR_FAIL(u_vec);
    geom(c, c->vec[u_vec].data(), GEOM_F);
    const int64_t ngp = (int64_t)c->ne * c->nGP;
    const int dm = c->dm, dd = dm * dm;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < ngp; ++t) {
        if (dm == 3) {
            double F[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) F[i][j] = c->F[t * dd + i * 3 + j];
            c->energy[t] = energy_density<3>(c->mat_kind, c->C, c->params[0], c->params[1], F);
        } else {
            double F[2][2];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) F[i][j] = c->F[t * dd + i * 2 + j];
            c->energy[t] = energy_density<2>(c->mat_kind, c->C, c->params[0], c->params[1], F);
        }
    }
    double s = 0.0;
    for (int64_t t = 0; t < ngp; ++t) s += c->energy[t] * c->vol[t];   // get_elasEng_kernel (:597-606)
    *total = s;
    return FEMCY_OK;
}

int femcy_extrapolate(femcy_ctx* ctx, int gp_field, int comp, const double* E, double* out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_element && E && out, "element tables not set or null arguments");
    const double* field = nullptr;
    int width = 1;
    switch (gp_field) {
        case FEMCY_GP_VOL: field = c->vol.data(); break;
        case FEMCY_GP_MISES: field = c->mises.data(); break;
        case FEMCY_GP_ENERGY: field = c->energy.data(); break;
        case FEMCY_GP_F: field = c->F.data(); width = c->dm * c->dm; break;
        case FEMCY_GP_SIGMA: field = c->sigma.data(); width = c->dm * c->dm; break;
        case FEMCY_GP_STRAIN: field = c->strain.data(); width = c->dm * c->dm; break;
        default: set_error("field %d cannot be extrapolated", gp_field); return FEMCY_EINVAL;
    }
    REQUIRE(comp >= 0 && comp < width, "component %d out of range for field %d", comp, gp_field);
    for (int64_t e = 0; e < c->ne; ++e)
        for (int a = 0; a < c->npe; ++a) {
            double acc = 0.0;
            for (int g = 0; g < c->nGP; ++g) acc += E[a * c->nGP + g] * field[(e * c->nGP + g) * width + comp];
            out[e * c->npe + a] = acc;
        }
    return FEMCY_OK;
}

// ---------------------------------------------------------------------------------- inspection
int femcy_get_K_ell(femcy_ctx* ctx, int32_t* ij, double* A) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern && ij && A, "pattern not built or null outputs");
    const int dm = c->dm, dd = dm * dm, W = c->max_row_blocks * dm;
    for (int32_t a = 0; a < c->nn; ++a) {
        const int L = (int)(c->rowptr[a + 1] - c->rowptr[a]);
        for (int r = 0; r < dm; ++r) {
            const int64_t i = (int64_t)a * dm + r;
            int32_t* row_ij = ij + i * (W + 1);
            double* row_A = A + i * W;
            row_ij[0] = L * dm;
            for (int t = 0; t < W; ++t) {
                row_ij[t + 1] = -1;
                row_A[t] = 0.0;
            }
            for (int j = 0; j < L; ++j)
                for (int cc = 0; cc < dm; ++cc) {
                    row_ij[1 + j * dm + cc] = c->col[c->rowptr[a] + j] * dm + cc;
                    row_A[j * dm + cc] = c->K[(c->rowptr[a] + j) * dd + r * dm + cc];
                }
        }
    }
    return FEMCY_OK;
}

int femcy_get_K_bsr(femcy_ctx* ctx, int32_t* rowptr, int32_t* colidx, double* out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_pattern && rowptr && colidx && out, "pattern not built or null outputs");
    const int dd = c->dm * c->dm;
    int64_t w = 0;
    rowptr[0] = 0;
    std::vector<std::pair<int32_t, int64_t>> order;
    for (int32_t a = 0; a < c->nn; ++a) {
        order.clear();
        for (int64_t p = c->rowptr[a]; p < c->rowptr[a + 1]; ++p) order.push_back({c->col[p], p});
        std::sort(order.begin(), order.end());
        for (auto& pr : order) {
            colidx[w] = pr.first;
            for (int k = 0; k < dd; ++k) out[w * dd + k] = c->K[pr.second * dd + k];
            ++w;
        }
        rowptr[a + 1] = (int32_t)w;
    }
    return FEMCY_OK;
}

int femcy_get_gp_field(femcy_ctx* ctx, int which, double* out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(c->have_element && out, "element tables not set or null output");
    const std::vector<double>* src = nullptr;
    switch (which) {
        case FEMCY_GP_DSDX: src = &c->dsdx; break;
        case FEMCY_GP_VOL: src = &c->vol; break;
        case FEMCY_GP_F: src = &c->F; break;
        case FEMCY_GP_SIGMA: src = &c->sigma; break;
        case FEMCY_GP_STRAIN: src = &c->strain; break;
        case FEMCY_GP_MISES: src = &c->mises; break;
        case FEMCY_GP_ENERGY: src = &c->energy; break;
        default: set_error("unknown Gauss-point field %d", which); return FEMCY_EINVAL;
    }
    std::memcpy(out, src->data(), src->size() * sizeof(double));
    return FEMCY_OK;
}

int femcy_timing(femcy_ctx* ctx, femcy_timing_t* out) {
    CTX_OR_FAIL(ctx);
    REQUIRE(out, "null output");
    *out = c->timing;
    return FEMCY_OK;
}
int femcy_timing_reset(femcy_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    c->timing = femcy_timing_t{};
    return FEMCY_OK;
}

// ------------------------------------------------------------ not on the host: device probes, multi-rank
int femcy_probe_stream(femcy_ctx*, int64_t, int32_t, int32_t, double*, int64_t*) {
    set_error("femcy_probe_stream measures a GPU: not available in the CPU backend");
    return FEMCY_EINVAL;
}
int femcy_probe_spmv(femcy_ctx*, int32_t, int32_t, double*) {
    set_error("femcy_probe_spmv measures a GPU: not available in the CPU backend");
    return FEMCY_EINVAL;
}
int femcy_probe_exchange(femcy_ctx*, int32_t, int32_t, double*) {
    set_error("femcy_probe_exchange measures a GPU: not available in the CPU backend");
    return FEMCY_EINVAL;
}
int femcy_persist_streamed_bytes(femcy_ctx* ctx, int64_t* bytes) {
    CTX_OR_FAIL(ctx);
    (void)c;
    REQUIRE(bytes, "null output");
    *bytes = 0;
    return FEMCY_OK;
}
#define NO_COMM(name)                                                                             \
    set_error(name ": the CPU backend holds the whole mesh in one process (no communicator)"); \
    return FEMCY_ECOMM
int femcy_comm_unique_id(void*) { NO_COMM("femcy_comm_unique_id"); }
int femcy_comm_local_id(void*) { NO_COMM("femcy_comm_local_id"); }
int femcy_comm_shm_id(void*, int64_t) { NO_COMM("femcy_comm_shm_id"); }
int femcy_probe_mailbox(femcy_ctx*, int32_t, double*) { NO_COMM("femcy_probe_mailbox"); }
int femcy_comm_allgather_host(femcy_ctx*, const void*, int32_t, void*) { NO_COMM("femcy_comm_allgather_host"); }
int femcy_comm_init(femcy_ctx*, int32_t, int32_t, const void*, int32_t, const int32_t*, const int32_t*, int32_t,
                    const uint8_t*) { NO_COMM("femcy_comm_init"); }
int femcy_comm_set_neighbours(femcy_ctx*, int32_t, const int32_t*, const int32_t*, const int32_t*) { NO_COMM("femcy_comm_set_neighbours"); }
int femcy_comm_tune(femcy_ctx*, int32_t, int32_t*, double*) { NO_COMM("femcy_comm_tune"); }
int femcy_iface_sum(femcy_ctx*, int) { NO_COMM("femcy_iface_sum"); }
int femcy_comm_mailbox_export(femcy_ctx*, void*) { NO_COMM("femcy_comm_mailbox_export"); }
int femcy_comm_mailbox_import(femcy_ctx*, int32_t, const void*) { NO_COMM("femcy_comm_mailbox_import"); }
int femcy_comm_persist_agree(femcy_ctx*, int32_t*) { NO_COMM("femcy_comm_persist_agree"); }
int femcy_comm_info(femcy_ctx* ctx, int32_t* rank, int32_t* nranks, int64_t* n_global) {
    CTX_OR_FAIL(ctx);
    if (rank) *rank = 0;
    if (nranks) *nranks = 1;
    if (n_global) *n_global = c->n;
    return FEMCY_OK;
}

}  // extern "C"
