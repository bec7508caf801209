// Jacobi-PCG as ONE persistent launch for systems that fit one wavefront-task per SIMD (<= ~6e5 DOF on MI355X: the
// headline 1 M-element C3D4 configuration), single rank.
//
// Why: with three launches per iteration (kernels_pcg.hip) the product streams the whole matrix from the Infinity
// Cache every iteration (198 MB, 31 us) and the two vector kernels are latency-bound launches of 5.6 us each.  Here
//   * one workgroup per CU stays resident for the whole solve; wave w of XCD k owns up to SPW slices (64 rows each)
//     of that XCD's contiguous slice range, lane = row, and keeps x, r, d, M of its rows in REGISTERS -- no vector
//     kernel, no vector traffic except the d the other waves gather (written once, 8 n bytes per iteration);
//   * the first `lds_rows` block rows of every wave live in LDS for the whole solve (160 KB per CU = 40 MB of the
//     198 MB matrix), the rest is streamed as before;
//   * the three synchronisation points of the recurrence (d.Ad before alpha, r.M.r before beta, the new d before the
//     next product) are grid barriers: per-XCD arrival counters + one top counter, relaxed agent-scope atomics, data
//     exchanged with sc1 (write-through) stores and sc1 loads -- no fences (cdna_hip_programming.md Guideline 16, R1).
//     Measured (tools/micro/barrier_probe.hip): 2.3 us per barrier at 256 workgroups, against 1.7-1.9 us for a
//     kernel boundary plus the ramp of a new launch.
// Recurrence, preconditioner and stopping rule are those of pcg_solve / the reference
// (conjugateGradientSolver.py:103-127); partial sums are combined in a fixed order, so a solve is bit-reproducible.
// d is double-buffered by iteration parity (a wave may gather d_k while a faster one already publishes d_k+1), as are
// the partial arrays.  Every spin is bounded: a timeout ends the launch with state.done = 3 and the host falls back to
// the three-kernel loop.
#include <cmath>
#include "ctx.hpp"

namespace femcy {

namespace {

constexpr int PBS = 256;        // 4 waves per workgroup, one workgroup per CU
constexpr int PNX = 8;

struct PersistPcg {
    const int32_t* slice_len;
    const int64_t* slice_off;
    const int32_t* bcol;
    const int32_t* node_of;
    const double* vals;
    const double* b;
    const double* M;
    double* x;
    double* dbuf;         // [2][npad]
    double* part1;        // [2][G]      d.Ad partials
    double* part2;        // [2][2 G]    (r.M.r, max|r|) pairs
    unsigned int* xc;     // [8][32]     per-XCD arrival counters (one cache line apart)
    unsigned int* top;
    PcgState* st;
    XcdRanges xr;         // slice range of each XCD
    int32_t npad, maxit, lds_rows;
    double eps;
};

__device__ __forceinline__ void pst(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pwave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double pwave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double pabs(double r) {   // fmax() drops NaN; keep it visible
    const double a = fabs(r);
    return (a != a) ? INFINITY : a;
}

// all workgroups of the launch; `round` counts the barriers since the launch (0, 1, 2, ...)
__device__ __forceinline__ bool grid_barrier(const PersistPcg& a, unsigned round, int* s_fail) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's sc1 stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned G = gridDim.x, k = blockIdx.x % PNX, members = G / PNX;
        const unsigned prev = __hip_atomic_fetch_add(a.xc + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == members * (round + 1))                   // last arrival of this XCD group in this round
            __hip_atomic_fetch_add(a.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(a.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < PNX * (round + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 25)) {
                *s_fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *s_fail == 0;
}

template <int DM, int SPW>
__global__ void __launch_bounds__(PBS) k_pcg_persist(PersistPcg a) {
    constexpr int DD = DM * DM, NP = DD / 2;
    extern __shared__ __attribute__((aligned(16))) char lds_persist[];
    __shared__ double sm1[PBS / 64], sm2[PBS / 64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    if (tid == 0) s_fail = 0;
    // wave's resident block rows: [q][k][lane] doubles, then [q][lane] columns
    double* lvals = reinterpret_cast<double*>(lds_persist) + (size_t)wave * a.lds_rows * DD * 64;
    int32_t* lcols = reinterpret_cast<int32_t*>(lds_persist + (size_t)4 * a.lds_rows * DD * 64 * 8) + (size_t)wave * a.lds_rows * 64;

    // ---- the wave's slices: XCD k (= blockIdx % 8, observed; speed only) owns the slice range xr[k] .. xr[k+1]
    const int xk = blockIdx.x % PNX;
    const int nwx = (G / PNX) * 4;                               // waves per XCD
    const int wx = (blockIdx.x / PNX) * 4 + wave;                // this wave among them
    int32_t sl[SPW], Ls[SPW], node[SPW];
    int64_t offs[SPW];
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
        const int32_t s = a.xr.start[xk] + wx + t * nwx;
        const bool act = s < a.xr.start[xk + 1];
        sl[t] = act ? s : -1;
        Ls[t] = act ? a.slice_len[s] : 0;
        offs[t] = act ? a.slice_off[s] : 0;
        node[t] = act ? a.node_of[(int64_t)s * 64 + lane] : -1;
    }
    // ---- resident part of the matrix -> LDS (once per solve)
    {
        int q = 0;
#pragma unroll
        for (int t = 0; t < SPW; ++t)
            for (int32_t j = 0; j < Ls[t] && q < a.lds_rows; ++j, ++q) {
                const double* src = a.vals + (offs[t] + j) * (int64_t)(DD * 64);
#pragma unroll
                for (int k = 0; k < DD; ++k) lvals[(q * DD + k) * 64 + lane] = src[kv_index<DM>(0, k, lane)];
                lcols[q * 64 + lane] = a.bcol[(offs[t] + j) * 64 + lane];
            }
    }
    // ---- x0 = 0, r = b, d = M r
    double xo[SPW][DM], rr[SPW][DM], mm[SPW][DM], dd[SPW][DM], Ad[SPW][DM];
    double accs = 0.0, accm = 0.0;
#pragma unroll
    for (int t = 0; t < SPW; ++t)
#pragma unroll
        for (int c = 0; c < DM; ++c) {
            const bool in = node[t] >= 0;
            const int64_t i = in ? (int64_t)node[t] * DM + c : 0;
            const double bi = in ? a.b[i] : 0.0, mi = in ? a.M[i] : 0.0;
            xo[t][c] = 0.0;
            rr[t][c] = bi;
            mm[t][c] = mi;
            dd[t][c] = mi * bi;
            Ad[t][c] = 0.0;
            if (in) pst(a.dbuf + i, dd[t][c]);
            accs += bi * mi * bi;
            accm = fmax(accm, pabs(bi));
        }
    unsigned round = 0;
    auto reduce_pair_publish = [&](double s, double m, double* slot2) {
        s = pwave_sum(s);
        m = pwave_max(m);
        if (lane == 0) {
            sm1[wave] = s;
            sm2[wave] = m;
        }
        __syncthreads();
        if (tid == 0) {
            pst(slot2, (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]));
            pst(slot2 + 1, fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3])));
        }
    };
    auto gather_pairs = [&](const double* base, double& s, double& m) {     // every workgroup: the same order
        double ps = 0.0, pm = 0.0;
        for (int k = tid; k < G; k += PBS) {
            ps += pld(base + 2 * k);
            pm = fmax(pm, pld(base + 2 * k + 1));
        }
        ps = pwave_sum(ps);
        pm = pwave_max(pm);
        __syncthreads();                                                     // sm1 / sm2 free again
        if (lane == 0) {
            sm1[wave] = ps;
            sm2[wave] = pm;
        }
        __syncthreads();
        s = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
        m = fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3]));
    };
    reduce_pair_publish(accs, accm, a.part2 + 2 * blockIdx.x);
    bool ok = grid_barrier(a, round++, &s_fail);
    double rMr = 0.0, r0 = 0.0;
    gather_pairs(a.part2, rMr, r0);
    double rmax = r0;
    int done = !ok ? 3 : ((r0 == 0.0) ? 1 : ((r0 != r0 || isinf(r0)) ? 2 : 0));
    int it = 0;
    while (!done && it < a.maxit) {
        __syncthreads();                                                     // sm1 / sm2 of the previous phase are read
        const double* dcur = a.dbuf + (size_t)(it & 1) * a.npad;
        // ---- Ad = K d for the wave's rows, d gathered with sc1 loads (the other XCDs wrote it with sc1 stores)
        double dot = 0.0;
        int q = 0;
#pragma unroll
        for (int t = 0; t < SPW; ++t) {
            double acc[DM];
#pragma unroll
            for (int r = 0; r < DM; ++r) acc[r] = 0.0;
            const int32_t L = Ls[t];
            int32_t j = 0;
            for (; j < L && q < a.lds_rows; ++j, ++q) {                       // resident block rows
                const int32_t col = lcols[q * 64 + lane];
                double xv[DM];
#pragma unroll
                for (int cc = 0; cc < DM; ++cc) xv[cc] = pld(dcur + (int64_t)col * DM + cc);
#pragma unroll
                for (int r = 0; r < DM; ++r)
#pragma unroll
                    for (int cc = 0; cc < DM; ++cc) acc[r] += lvals[(q * DD + r * DM + cc) * 64 + lane] * xv[cc];
            }
            const int32_t* __restrict__ bc = a.bcol + offs[t] * 64 + lane;
            const double2* __restrict__ vp = reinterpret_cast<const double2*>(a.vals + offs[t] * (int64_t)(DD * 64)) + lane;
            const double* __restrict__ vs = a.vals + offs[t] * (int64_t)(DD * 64) + NP * 128 + lane;
#pragma unroll 4
            for (; j < L; ++j) {                                             // streamed block rows
                const int32_t col = bc[(int64_t)j * 64];
                double xv[DM], e[DD];
#pragma unroll
                for (int kp = 0; kp < NP; ++kp) {
                    const double2 v2 = vp[(int64_t)j * (DD * 32) + kp * 64];
                    e[2 * kp] = v2.x;
                    e[2 * kp + 1] = v2.y;
                }
                if (DD & 1) e[DD - 1] = vs[(int64_t)j * (DD * 64)];
#pragma unroll
                for (int cc = 0; cc < DM; ++cc) xv[cc] = pld(dcur + (int64_t)col * DM + cc);
#pragma unroll
                for (int r = 0; r < DM; ++r)
#pragma unroll
                    for (int cc = 0; cc < DM; ++cc) acc[r] += e[r * DM + cc] * xv[cc];
            }
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                Ad[t][r] = acc[r];
                if (node[t] >= 0) dot += dd[t][r] * acc[r];
            }
        }
        dot = pwave_sum(dot);
        if (lane == 0) sm1[wave] = dot;
        __syncthreads();
        if (tid == 0) pst(a.part1 + (size_t)(it & 1) * G + blockIdx.x, (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]));
        if (!grid_barrier(a, round++, &s_fail)) { done = 3; break; }
        // ---- alpha; x, r; partials of (r.M.r, max|r|)
        double ps = 0.0;
        for (int k = tid; k < G; k += PBS) ps += pld(a.part1 + (size_t)(it & 1) * G + k);
        ps = pwave_sum(ps);
        __syncthreads();
        if (lane == 0) sm1[wave] = ps;
        __syncthreads();
        const double dAd = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
        const double alpha = rMr / dAd;
        accs = 0.0;
        accm = 0.0;
#pragma unroll
        for (int t = 0; t < SPW; ++t)
#pragma unroll
            for (int c = 0; c < DM; ++c) {
                xo[t][c] += alpha * dd[t][c];
                const double ri = rr[t][c] - alpha * Ad[t][c];
                rr[t][c] = ri;
                if (node[t] >= 0) {
                    accs += ri * mm[t][c] * ri;
                    accm = fmax(accm, pabs(ri));
                }
            }
        __syncthreads();
        reduce_pair_publish(accs, accm, a.part2 + (size_t)((it + 1) & 1) * 2 * G + 2 * blockIdx.x);
        if (!grid_barrier(a, round++, &s_fail)) { done = 3; break; }
        double rMr_new = 0.0;
        gather_pairs(a.part2 + (size_t)((it + 1) & 1) * 2 * G, rMr_new, rmax);
        ++it;
        if (rmax != rmax || isinf(rmax) || rMr_new != rMr_new) {
            done = 2;
        } else if (rmax < a.eps * r0) {
            done = 1;
        } else {
            // ---- d = M r + beta d, published for the next product
            const double beta = rMr_new / rMr;
            double* dnext = a.dbuf + (size_t)(it & 1) * a.npad;
#pragma unroll
            for (int t = 0; t < SPW; ++t)
#pragma unroll
                for (int c = 0; c < DM; ++c) {
                    dd[t][c] = mm[t][c] * rr[t][c] + beta * dd[t][c];
                    if (node[t] >= 0) pst(dnext + (int64_t)node[t] * DM + c, dd[t][c]);
                }
            rMr = rMr_new;
            if (it < a.maxit && !grid_barrier(a, round++, &s_fail)) { done = 3; break; }
        }
        if (done) rMr = rMr_new;
    }
#pragma unroll
    for (int t = 0; t < SPW; ++t)
        if (node[t] >= 0) {
#pragma unroll
            for (int c = 0; c < DM; ++c) a.x[(int64_t)node[t] * DM + c] = xo[t][c];
        }
    if (blockIdx.x == 0 && tid == 0) {
        a.st->iters = it;
        a.st->r0 = r0;
        a.st->rmax = rmax;
        a.st->done = done;
        a.st->rMr[0] = rMr;
    }
}

}  // namespace

// eligibility + launch; *handled = false when the system does not qualify (too small, too large, multi-rank)
int pcg_persist_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, bool* handled) {
    *handled = false;
    constexpr int SPW = 3;
    if (!c->opt_persist || c->comm || c->persist_failed) return FEMCY_OK;
    const int G = (c->persist_cus / PNX) * PNX;                  // one workgroup per CU
    if (G < PNX) return FEMCY_OK;
    const int nwx = (G / PNX) * 4;
    int32_t maxrange = 0;
    for (int k = 0; k < PNX; ++k) maxrange = std::max(maxrange, c->xcd.start[k + 1] - c->xcd.start[k]);
    if (maxrange > SPW * nwx || c->nslices < G) return FEMCY_OK;  // does not fit / too small to be worth a whole chip
    const int64_t npad = (c->n + 1) & ~(int64_t)1;
    const int DD = c->dm * c->dm;
    int lds_rows = c->opt_persist_lds < 0 ? (int)((c->small_max_lds - 2048) / (4 * 64 * (DD * 8 + 4))) : c->opt_persist_lds;
    lds_rows = std::max(0, std::min(lds_rows, SPW * (int)c->max_row_blocks));
    const size_t lds = (size_t)4 * lds_rows * 64 * (DD * 8 + 4) + 16;
    const int64_t need = 2 * npad + 2 * G + 4 * G + 160;          // + 8 x 32 + 32 barrier counters (4 bytes each)
    if (!c->d_persist || c->persist_cap < need) {
        if (c->d_persist) (void)hipFree(c->d_persist);
        c->d_persist = nullptr;
        c->persist_cap = need;
        FEMCY_HIP(hipMalloc((void**)&c->d_persist, sizeof(double) * need));
    }
    PersistPcg a;
    a.slice_len = c->d_slice_len; a.slice_off = c->d_slice_off; a.bcol = c->d_bcol; a.node_of = c->d_node_of;
    a.vals = c->d_Kvals; a.b = d_b; a.M = c->d_M; a.x = d_x;
    a.dbuf = c->d_persist;
    a.part1 = c->d_persist + 2 * npad;
    a.part2 = a.part1 + 2 * G;
    a.xc = reinterpret_cast<unsigned int*>(a.part2 + 4 * G);
    a.top = a.xc + 8 * 32;
    a.st = c->d_state;
    a.xr = c->xcd;
    a.npad = (int32_t)npad; a.maxit = maxit; a.lds_rows = lds_rows; a.eps = eps;
    FEMCY_HIP(hipMemsetAsync(a.xc, 0, sizeof(unsigned int) * (8 * 32 + 32), c->stream));
#define FEMCY_PERSIST(DM_)                                                                                        \
    do {                                                                                                          \
        const void* fn = reinterpret_cast<const void*>(&k_pcg_persist<DM_, SPW>);                                 \
        if (lds > 48 * 1024)                                                                                      \
            FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_pcg_persist<DM_, SPW>), dim3(G), dim3(PBS), lds, c->stream, a);                     \
    } while (0)
    if (c->dm == 3) FEMCY_PERSIST(3); else FEMCY_PERSIST(2);
#undef FEMCY_PERSIST
    FEMCY_HIP(hipGetLastError());
    *handled = true;
    return FEMCY_OK;
}

}  // namespace femcy
