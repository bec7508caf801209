// Jacobi-PCG as ONE persistent launch for single-rank systems that fit one wavefront-task per SIMD (<= 4 slices of
// 64 rows per wave: ~7e5 DOF on MI355X) and whose matrix streams from the Infinity Cache (<= 240 MiB stored): the
// headline 1 M-element C3D4 configuration.  DESIGN.md section 3 has the measurements.
//
// Why: with three launches per iteration (kernels_pcg.hip) the product streams the whole matrix from the Infinity
// Cache every iteration (198 MB, 31 us) and the two vector kernels are latency-bound launches of 5.6 us each.  Here
//   * one workgroup of four waves per CU stays resident for the whole solve (one wave per SIMD = 512 registers per
//     lane); a wave owns up to SPW slices of its XCD's contiguous slice range (handed out longest first, host side),
//     lane = row, and keeps x, r, d, M, Ad of its rows in REGISTERS -- no vector kernel, no vector traffic except the
//     d the other waves gather (written once, 8 n bytes per iteration);
//   * block rows 0 .. RJ-1 of every slice live in registers (AGPRs) and the next `lds_rows` block rows of the wave in
//     LDS for the whole solve (47 % of the 198 MB matrix at RJ = 4); the first streamed batch of the next product is
//     requested right after the current one and arrives during the barrier waits; the rest is streamed in batches
//     of CH block rows (columns, values, then the gathers as sc1 buffer loads, then the multiplies);
//   * the three synchronisation points of the recurrence (d.Ad before alpha, r.M.r before beta, the new d before the
//     next product) are grid barriers: per-XCD arrival counters + one top counter, relaxed agent-scope atomics, data
//     exchanged with sc1 (write-through) stores and sc1 loads -- no fences (cdna_hip_programming.md Guideline 16, R1).
//     Measured (tools/micro/barrier_probe.hip): 2.2-2.4 us per barrier at 256 workgroups, against 1.7-1.9 us for a
//     kernel boundary plus the ramp of a new launch.
// Recurrence, preconditioner and stopping rule are those of pcg_solve / the reference
// (conjugateGradientSolver.py:103-127); partial sums are combined in a fixed order, so a solve is bit-reproducible.
// d is double-buffered by iteration parity (a wave may gather d_k while a faster one already publishes d_k+1), as are
// the partial arrays.  Every spin is bounded: the workgroup that times out poisons the top counter (which releases
// all others), the launch ends with state.done = 3 and the host falls back to the three-kernel loop.
#include <algorithm>
#include <cmath>
#include <vector>
#include "ctx.hpp"
#include "wave_reduce.hpp"

namespace femcy {

namespace {

constexpr int PBS = 256;        // 4 waves per workgroup, one workgroup per CU
constexpr int PNX = 8;
constexpr int CH = 4;         // block rows per batch of the LDS-resident and the streamed part

struct PersistPcg {
    const int32_t* slice_len;
    const int64_t* slice_off;
    const int32_t* bcol;
    const int32_t* node_of;
    const int32_t* assign;  // [8 * waves per XCD][SPW] slices of each wave, -1 = none
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
    int32_t npad, maxit, lds_rows, dbg;
    double eps;
};

__device__ __forceinline__ void pst(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pld(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pabs(double r) {   // fmax() drops NaN; keep it visible
    const double a = fabs(r);
    return (a != a) ? INFINITY : a;
}

// all workgroups of the launch; `round` counts the barriers since the launch (0, 1, 2, ...)
// `mid` runs in every thread between the arrival and the wait: loads issued there travel while the workgroup waits
// (issued before the arrival they would sit in front of it in the in-order return queue: s_waitcnt vmcnt(0))
template <class Mid>
__device__ __forceinline__ bool grid_barrier(const PersistPcg& a, unsigned round, int* s_fail, Mid&& mid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's sc1 stores have left
    __syncthreads();
    const unsigned G = gridDim.x, k = blockIdx.x % PNX, members = G / PNX;
    unsigned prev = 0;
    if (threadIdx.x == 0)
        prev = __hip_atomic_fetch_add(a.xc + 32 * k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mid();
    if (threadIdx.x == 0) {
        if (prev + 1 == members * (round + 1))                   // last arrival of this XCD group in this round
            __hip_atomic_fetch_add(a.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // bounded wait (a legitimate one is < 100 us): on a time-out the top counter is poisoned, which releases every
        // other workgroup -- those spinning now and those that reach a barrier later -- with the same verdict
        constexpr unsigned POISON = 0x80000000u;
        unsigned spins = 0;
        while (!(a.dbg & 8)) {
            const unsigned v = __hip_atomic_load(a.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v & POISON) {
                *s_fail = 1;
                break;
            }
            if (v >= PNX * (round + 1)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) {
                __hip_atomic_fetch_or(a.top, POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_fail = 1;
                break;
            }
        }
    }
    __syncthreads();
    return *s_fail == 0;
}
__device__ __forceinline__ bool grid_barrier(const PersistPcg& a, unsigned round, int* s_fail) {
    return grid_barrier(a, round, s_fail, [] {});
}

template <int DM, int SPW, int RJ>
__global__ void __launch_bounds__(PBS) k_pcg_persist(PersistPcg a) {
    constexpr int DD = DM * DM, NP = DD / 2;
    extern __shared__ __attribute__((aligned(16))) char lds_persist[];
    __shared__ double sm1[PBS / 64], sm2[PBS / 64];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and known to be so: scalar registers
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
        const int32_t s = __builtin_amdgcn_readfirstlane(a.assign[((size_t)xk * nwx + wx) * SPW + t]);
        const bool act = s >= 0;
        sl[t] = act ? s : -1;
        Ls[t] = act ? __builtin_amdgcn_readfirstlane(a.slice_len[s]) : 0;
        const int64_t o = act ? a.slice_off[s] : 0;
        offs[t] = ((int64_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o);
        node[t] = act ? a.node_of[(int64_t)s * 64 + lane] : -1;
    }
    // ---- resident part of the matrix (once per solve): block rows 0 .. RJ-1 of every slice -> registers (a row the
    // slice does not have is a zero block on the lane's own node), the next lds_rows block rows of the wave -> LDS
    double rv[SPW][RJ > 0 ? RJ : 1][DD];
    int32_t rcl[SPW][RJ > 0 ? RJ : 1];
#pragma unroll
    for (int t = 0; t < SPW; ++t)
#pragma unroll
        for (int jj = 0; jj < RJ; ++jj) {
            const bool has = jj < Ls[t];
            const double* src = a.vals + (offs[t] + (has ? jj : 0)) * (int64_t)(DD * 64);
#pragma unroll
            for (int k = 0; k < DD; ++k) rv[t][jj][k] = has ? src[kv_index<DM>(0, k, lane)] : 0.0;
            rcl[t][jj] = has ? a.bcol[(offs[t] + jj) * 64 + lane] : (node[t] >= 0 ? node[t] : 0);
        }
    // LDS rows are handed out from the LAST slice backwards, so that slice 0 keeps streamed rows: its first batch is
    // the one prefetched during the synchronisation windows (below).  jl[t] .. je[t]-1 = the slice's rows in LDS.
    int32_t jl[SPW], je[SPW], ql[SPW];
    {
        int qn = 0;
#pragma unroll
        for (int t = SPW - 1; t >= 0; --t) {
            jl[t] = min(RJ, Ls[t]);
            const int nl = max(0, min(Ls[t] - jl[t], a.lds_rows - qn));
            ql[t] = qn;
            je[t] = jl[t] + nl;
            qn += nl;
            for (int32_t j = jl[t]; j < je[t]; ++j) {
                const int q = ql[t] + (j - jl[t]);
                const double* src = a.vals + (offs[t] + j) * (int64_t)(DD * 64);
#pragma unroll
                for (int k = 0; k < DD; ++k) lvals[(q * DD + k) * 64 + lane] = src[kv_index<DM>(0, k, lane)];
                lcols[q * 64 + lane] = a.bcol[(offs[t] + j) * 64 + lane];
            }
        }
    }
    // d is gathered with sc1 BUFFER loads: the same cache policy as an agent-scope atomic load (the other XCDs wrote d
    // with sc1 stores), but an ordinary load to the compiler, which may then issue the gathers of several block rows
    // before the first wait (with atomic loads it serialised them: 15 + 8 dependent L2 round trips per product)
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dbuf, 0, (int)((size_t)2 * a.npad * sizeof(double)), 0x00020000);
    auto gather_d = [&](int32_t col, int32_t parity_off, double (&xv)[DM]) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int cc = 0; cc < DM; ++cc) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(drsrc, (col * DM + cc) * 8, parity_off, 16 /* sc1 */);
            xv[cc] = __hiloint2double((int)w.y, (int)w.x);
        }
    };
    // nb (<= CH) consecutive block rows of a slice: columns and values (rows beyond nb: the column of the last one,
    // so that its gather stays in range; no values)
    auto load_rows = [&](const int32_t* __restrict__ bc, const double2* __restrict__ vp, const double* __restrict__ vs,
                         int32_t j, int nb, int32_t (&col)[CH], double (&e)[CH][DD]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) col[u] = bc[(int64_t)(j + max(0, min(u, nb - 1))) * 64];
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (u < nb) {
#pragma unroll
                for (int kp = 0; kp < NP; ++kp) {
                    const double2 v2 = vp[(int64_t)(j + u) * (DD * 32) + kp * 64];
                    e[u][2 * kp] = v2.x;
                    e[u][2 * kp + 1] = v2.y;
                }
                if (DD & 1) e[u][DD - 1] = vs[(int64_t)(j + u) * (DD * 64)];
            } else {
                // defined on every path: a conditionally rewritten loop-carried buffer would keep its old value alive
                // through the whole iteration (the prefetch buffer then costs 76 registers at the product's peak)
#pragma unroll
                for (int k = 0; k < DD; ++k) e[u][k] = 0.0;
            }
    };
    const int32_t* __restrict__ bc0 = a.bcol + offs[0] * 64 + lane;
    const double2* __restrict__ vp0 = reinterpret_cast<const double2*>(a.vals + offs[0] * (int64_t)(DD * 64)) + lane;
    const double* __restrict__ vs0 = a.vals + offs[0] * (int64_t)(DD * 64) + NP * 128 + lane;
    const int npf = (a.dbg & 16) ? 0 : max(0, min(CH, Ls[0] - je[0]));   // rows of the prefetched batch
    int32_t pcol[CH];
    double pe[CH][DD];
    const int32_t jpf = npf > 0 ? je[0] : 0;                               // (npf = 0: row 0's column, unused)
    load_rows(bc0, vp0, vs0, jpf, npf, pcol, pe);
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
        s = wave_sum(s);
        m = wave_max(m);
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
        ps = wave_sum(ps);
        pm = wave_max(pm);
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
        const int32_t poff = (it & 1) * a.npad * 8;                           // byte offset of this iteration's d
        // ---- Ad = K d for the wave's rows, d gathered with sc1 loads (the other XCDs wrote it with sc1 stores)
        double dot = 0.0;
#pragma unroll
        for (int t = 0; t < SPW; ++t) {
            double acc[DM];
#pragma unroll
            for (int r = 0; r < DM; ++r) acc[r] = 0.0;
            const int32_t L = Ls[t];
            const int32_t* __restrict__ bc = a.bcol + offs[t] * 64 + lane;
            const double2* __restrict__ vp = reinterpret_cast<const double2*>(a.vals + offs[t] * (int64_t)(DD * 64)) + lane;
            const double* __restrict__ vs = a.vals + offs[t] * (int64_t)(DD * 64) + NP * 128 + lane;
            int32_t j = je[t];                                               // first streamed block row
            // slice 0: its first streamed batch was loaded while the wave sat in the last synchronisation points
            if (t == 0 && npf > 0 && !(a.dbg & 1)) {
                double xg[CH][DM];
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(pcol[u], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < npf) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc) acc[r] += pe[u][r * DM + cc] * xg[u][cc];
                    }
                j += npf;
            }
            // block rows held in registers: their gathers are issued in batches before the first multiply
            if (RJ > 0 && !(a.dbg & 4)) {
                constexpr int RB = RJ > 4 ? 3 : (RJ > 0 ? RJ : 1);               // rows per batch (register budget)
#pragma unroll
                for (int j0 = 0; j0 < RJ; j0 += RB) {
                    double xg[RB][DM];
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        if (j0 + u < RJ) gather_d(rcl[t][j0 + u], poff, xg[u]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        if (j0 + u < RJ) {
#pragma unroll
                            for (int r = 0; r < DM; ++r)
#pragma unroll
                                for (int cc = 0; cc < DM; ++cc) acc[r] += rv[t][j0 + u][r * DM + cc] * xg[u][cc];
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // block rows held in LDS, CH at a time (a short batch repeats its last row's gather and skips the multiply)
            for (int32_t jr = jl[t]; jr < je[t] && !(a.dbg & 2); jr += CH) {
                const int nb = min(CH, je[t] - jr);
                const int q = ql[t] + (jr - jl[t]);
                double xg[CH][DM];
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(lcols[(q + min(u, nb - 1)) * 64 + lane], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < nb) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc)
                                acc[r] += lvals[((q + u) * DD + r * DM + cc) * 64 + lane] * xg[u][cc];
                    }
            }
            // streamed block rows, CH at a time: columns, then the values, then the gathers, then the multiplies
            while (j < L && !(a.dbg & 1)) {
                const int nb = min(CH, L - j);
                int32_t col[CH];
                double e[CH][DD], xg[CH][DM];
                load_rows(bc, vp, vs, j, nb, col, e);
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(col[u], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < nb) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc) acc[r] += e[u][r * DM + cc] * xg[u][cc];
                    }
                j += nb;
            }
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                Ad[t][r] = acc[r];
                if (node[t] >= 0) dot += dd[t][r] * acc[r];
            }
        }
        // the matrix does not change: slice 0's first streamed batch for the NEXT product is requested inside the first
        // barrier (after the arrival, so that it does not delay it) and arrives while the wave waits in the three
        // synchronisation points (registers and memory system are idle there)
        dot = wave_sum(dot);
        if (lane == 0) sm1[wave] = dot;
        __syncthreads();
        if (tid == 0) pst(a.part1 + (size_t)(it & 1) * G + blockIdx.x, (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]));
        if (!grid_barrier(a, round++, &s_fail, [&] { load_rows(bc0, vp0, vs0, jpf, npf, pcol, pe); })) { done = 3; break; }
        // ---- alpha; x, r; partials of (r.M.r, max|r|)
        double ps = 0.0;
        for (int k = tid; k < G; k += PBS) ps += pld(a.part1 + (size_t)(it & 1) * G + k);
        ps = wave_sum(ps);
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
        if (a.dbg & 15) rmax = 1.0, rMr_new = 1.0;   // bits 0-3 skip work: keep iterating on whatever numbers result
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
    if (!c->opt_persist || c->comm || c->persist_failed) return FEMCY_OK;
    const int G = ((c->opt_persist_wgs > 0 ? c->opt_persist_wgs : c->persist_cus) / PNX) * PNX;   // one workgroup per CU
    if (G < PNX) return FEMCY_OK;
    const int nwx = (G / PNX) * 4;
    int32_t maxrange = 0;
    for (int k = 0; k < PNX; ++k) maxrange = std::max(maxrange, c->xcd.start[k + 1] - c->xcd.start[k]);
    // every wave gets its slices (<= 4), and the chip is filled 1.5 times over: below ~380 slices the 13 us of
    // synchronisation per iteration exceed the (graph-replayed) three-launch iteration (size sweep in DESIGN.md)
    if (maxrange > 4 * nwx || c->nslices < G || (c->nslices < G + G / 2 && c->opt_persist < 2)) return FEMCY_OK;
    const int64_t npad = (c->n + 1) & ~(int64_t)1;
    const int DD = c->dm * c->dm;
    int lds_rows = c->opt_persist_lds < 0 ? (int)((c->small_max_lds - 2048) / (4 * 64 * (DD * 8 + 4))) : c->opt_persist_lds;
    const int SPW = maxrange > 3 * nwx ? 4 : 3;                   // slices per wave (the kernel's register arrays)
    lds_rows = std::max(0, std::min(lds_rows, SPW * (int)c->max_row_blocks));
    // ... and the part of the matrix that is STREAMED every iteration has to come from the Infinity Cache (256 MiB):
    // with one wave per SIMD there is little latency hiding for HBM.  Measured: 1.4 M C3D4 elements (277 MB stored,
    // 100 MB of it resident) 41.7 us per iteration here against 63.1 with three launches; 124 k C3D10 (380 MB stored,
    // 280 MB streamed) 99-103 us here against 93.
    const int rj = c->dm == 3 ? (SPW == 3 ? c->opt_persist_rj : (c->opt_persist_rj ? 3 : 0)) : (c->opt_persist_rj ? 5 : 0);
    const int64_t row_bytes = (int64_t)(DD * 8 + 4) * 64;
    const int64_t kbytes = c->stored_rows * row_bytes;
    const int64_t resident = (int64_t)G * 4 * (SPW * rj + lds_rows) * row_bytes;   // upper bound (short slices hold less)
    if (kbytes - resident > c->persist_max_bytes && c->opt_persist < 2) return FEMCY_OK;
    const size_t lds = (size_t)4 * lds_rows * 64 * (DD * 8 + 4) + 16;
    const int64_t need = 2 * npad + 2 * G + 4 * G + 160;          // + 8 x 32 + 32 barrier counters (4 bytes each)
    if (!c->d_persist || c->persist_cap < need) {
        if (c->d_persist) (void)hipFree(c->d_persist);
        c->d_persist = nullptr;
        c->persist_cap = need;
        FEMCY_HIP(hipMalloc((void**)&c->d_persist, sizeof(double) * need));
    }
    // slices of each wave: XCD k's waves share the slice range xcd[k] .. xcd[k+1] (the ranges are balanced by stored
    // block rows); inside it the slices go longest first to the wave with the least rows so far (LPT) -- the sigma-
    // sorted windows would otherwise hand all long slices to the same waves
    {
        std::vector<int64_t> key = {G, SPW, c->pattern_serial};
        for (int k = 0; k <= PNX; ++k) key.push_back(c->xcd.start[k]);
        if (key != c->persist_assign_key || !c->d_persist_assign) {
            std::vector<int32_t> assign((size_t)PNX * nwx * SPW, -1);
            std::vector<int32_t> order, load(nwx), cnt(nwx);
            for (int k = 0; k < PNX; ++k) {
                order.clear();
                for (int32_t s = c->xcd.start[k]; s < c->xcd.start[k + 1]; ++s) order.push_back(s);
                std::stable_sort(order.begin(), order.end(),
                                 [&](int32_t x, int32_t y) { return c->h_slice_len[x] > c->h_slice_len[y]; });
                std::fill(load.begin(), load.end(), 0);
                std::fill(cnt.begin(), cnt.end(), 0);
                for (int32_t s : order) {
                    int best = -1;
                    for (int w = 0; w < nwx; ++w)
                        if (cnt[w] < SPW && (best < 0 || load[w] < load[best])) best = w;
                    assign[((size_t)k * nwx + best) * SPW + cnt[best]++] = s;
                    load[best] += c->h_slice_len[s];
                }
            }
            if (c->d_persist_assign) (void)hipFree(c->d_persist_assign);
            c->d_persist_assign = nullptr;
            FEMCY_HIP(hipMalloc((void**)&c->d_persist_assign, assign.size() * sizeof(int32_t)));
            FEMCY_HIP(hipMemcpyAsync(c->d_persist_assign, assign.data(), assign.size() * sizeof(int32_t),
                                     hipMemcpyHostToDevice, c->stream));
            FEMCY_HIP(hipStreamSynchronize(c->stream));
            c->persist_assign_key = key;
        }
    }
    PersistPcg a;
    a.assign = c->d_persist_assign;
    a.slice_len = c->d_slice_len; a.slice_off = c->d_slice_off; a.bcol = c->d_bcol; a.node_of = c->d_node_of;
    a.vals = c->d_Kvals; a.b = d_b; a.M = c->d_M; a.x = d_x;
    a.dbuf = c->d_persist;
    a.part1 = c->d_persist + 2 * npad;
    a.part2 = a.part1 + 2 * G;
    a.xc = reinterpret_cast<unsigned int*>(a.part2 + 4 * G);
    a.top = a.xc + 8 * 32;
    a.st = c->d_state;
    a.npad = (int32_t)npad; a.maxit = maxit; a.lds_rows = lds_rows; a.eps = eps; a.dbg = c->opt_persist_dbg;
    FEMCY_HIP(hipMemsetAsync(a.xc, 0, sizeof(unsigned int) * (8 * 32 + 32), c->stream));
#define FEMCY_PERSIST(DM_, SPW_, RJ_)                                                                             \
    do {                                                                                                          \
        const void* fn = reinterpret_cast<const void*>(&k_pcg_persist<DM_, SPW_, RJ_>);                           \
        if (lds > 48 * 1024)                                                                                      \
            FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_pcg_persist<DM_, SPW_, RJ_>), dim3(G), dim3(PBS), lds, c->stream, a);               \
    } while (0)
    const size_t tp = timing_begin(c, T_PERSIST);
    // register-resident block rows per slice: what 512 VGPRs per lane hold next to the vectors and the streaming
    // buffers (dm 3: 5 rows x 3 slices or 3 rows x 4 slices of 19 registers each)
    if (c->dm == 3 && SPW == 3) {
        if (c->opt_persist_rj == 5) FEMCY_PERSIST(3, 3, 5);
        else if (c->opt_persist_rj == 4) FEMCY_PERSIST(3, 3, 4);
        else FEMCY_PERSIST(3, 3, 0);
    } else if (c->dm == 3) {
        if (c->opt_persist_rj) FEMCY_PERSIST(3, 4, 3); else FEMCY_PERSIST(3, 4, 0);
    } else if (SPW == 3) {
        if (c->opt_persist_rj) FEMCY_PERSIST(2, 3, 5); else FEMCY_PERSIST(2, 3, 0);
    } else {
        if (c->opt_persist_rj) FEMCY_PERSIST(2, 4, 5); else FEMCY_PERSIST(2, 4, 0);
    }
#undef FEMCY_PERSIST
    timing_end(c, tp);
    FEMCY_HIP(hipGetLastError());
    *handled = true;
    return FEMCY_OK;
}

}  // namespace femcy
