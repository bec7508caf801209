// Jacobi-preconditioned CG on the blocked SELL-64 matrix, plus the axpy-class vector helpers.
//
// Reference: conjugateGradientSolver.py:10-127 (ConjugateGradientSolver_rowMajor) and
// tiGadgets.py:5-37,67-70.  The reference issues 8 kernels and 4 blocking device->host scalar reads
// per iteration; here one iteration is 3 kernels and no host round trip:
//
//   k_spmv        Ad = K d, per-block partials of d.Ad                      (compute_Ad + dot_product)
//   k_update_xr   alpha = rMr / sum(partials);  x += alpha d;  r -= alpha Ad;
//                 per-block partials of r.M.r and max|r|                    (update_x, update_r,
//                                                                            compute_rMr, rmax)
//   k_update_d    beta = rMr_new / rMr;  d = M r + beta d;  block 0 publishes rMr_new, rmax, iters
//                 and the converged flag                                    (update_d + stopping rule)
//
// Scalars never leave the device: every block of the consuming kernel re-reduces the producer's
// partials in a fixed order (deterministic, no atomics, no extra launch); the kernel boundary is
// the release/acquire.  The recurrence, the initial guess x0 = 0, M = 1/diag(K) and the stopping
// rule max|r| < eps * max|r0| are the reference's.  Once `done` is set the remaining queued kernels
// exit immediately, so x is exactly the iterate at which the reference would `break`; the host
// polls the flag every `opt_poll` iterations through pinned memory.
//
// SpMV layout: lane = node (dm rows), slice = 64 nodes = one wavefront, values stored
// [stored_row][k = r*dm+c][lane] so every load instruction of a wave is one contiguous 512-byte
// segment; the block-column index is stored once per block ([stored_row][lane]); x is gathered
// dm doubles at a time.  Workgroups are remapped so that each XCD (private 4 MiB L2) walks one
// contiguous range of slices and the x-gathers of neighbouring slices hit the same L2.
#include <cmath>
#include <type_traits>
#include <hip/hip_ext.h>
#include "ctx.hpp"
#include "wave_reduce.hpp"
#include "granule.hpp"

namespace femcy {

constexpr int BS = 256;
constexpr int NXCD = 8;

// ------------------------------------------------------------------------------------ reductions
// wave_sum / wave_max: wave_reduce.hpp (DPP network; the result is in every lane)
// result broadcast to every thread; sm must hold BS/64 doubles; two calls need distinct sm
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < BS / 64; ++i) t += sm[i];
    __syncthreads();
    return t;
}
__device__ __forceinline__ double block_max(double v, double* sm) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = sm[0];
#pragma unroll
    for (int i = 1; i < BS / 64; ++i) t = fmax(t, sm[i]);
    __syncthreads();
    return t;
}
__device__ __forceinline__ double reduce_partials_sum(const double* __restrict__ part, int np, double* sm) {
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += BS) v += part[i];
    return block_sum(v, sm);
}
__device__ __forceinline__ double reduce_partials_max(const double* __restrict__ part, int np, double* sm) {
    double v = 0.0;
    for (int i = threadIdx.x; i < np; i += BS) v = fmax(v, part[i]);
    return block_max(v, sm);
}
__device__ __forceinline__ double nan_to_inf_abs(double r) {   // fmax() drops NaN; keep it visible
    double a = fabs(r);
    return (a != a) ? INFINITY : a;
}

// the dm doubles of a column node are contiguous (8-byte aligned): ONE 16-byte load (+ one 8-byte load for dm = 3)
// instead of dm 8-byte loads -- a third fewer texture-address cycles on the gathers, which is what bounds the product
// once the matrix streams from HBM (C3D10, 8 M elements: TA_TA_BUSY 74 % in round 2).  Buffer loads: dword alignment
// suffices, out-of-range lanes (none here) would read 0.
typedef unsigned int spmv_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int spmv_u32x4 __attribute__((ext_vector_type(4)));
template <int DM>
__device__ __forceinline__ void gather_x(const __amdgpu_buffer_rsrc_t rs, int32_t col, double (&xv)[DM]) {
    const spmv_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, col * (DM * 8), 0, 0);
    xv[0] = __hiloint2double((int)w.y, (int)w.x);
    xv[1] = __hiloint2double((int)w.w, (int)w.z);
    if (DM == 3) {
        const spmv_u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(rs, col * (DM * 8) + 16, 0, 0);
        xv[DM - 1] = __hiloint2double((int)w2.y, (int)w2.x);
    }
}

// ------------------------------------------------------------------------------------------ SpMV
// position of workgroup `wgx` of an XCD inside round `rnd` of the XCD's task list (a permutation of 0 .. bpx-1 per round).
// W = tasks per sort window (rows are sorted by length inside windows of 64 slices): the workgroup's position INSIDE the
// window advances by `rot` from one round to the next whatever bpx mod W is; rot = 0: the same position every round
template <int W>
__device__ __forceinline__ int spmv_rot(int wgx, int rnd, int bpx, int rot) {
    if (rot == 0) return wgx;
    const int step = (((rot - bpx) % W) + W) % W;
    return (int)((unsigned)(wgx + (step * rnd) % W) % (unsigned)bpx);
}
// WPS = wavefronts per slice: long rows (C3D10: 27-65 blocks per node) are split into WPS contiguous j-chunks
// handled by WPS waves of the same workgroup and summed through LDS, so that the chain per wave stays short and
// the few thousand slices still fill 1024 SIMDs evenly.
// slice_list != nullptr: the XCD ranges count positions of that list instead of slices (multi-rank split product:
// interface slices and interior slices are two launches over the two halves of one list)
// node_of == nullptr: x and y are in storage order and bcol holds storage positions (nn = nslices * 64): lanes of a wave
// are rows of one length class in ascending order, their j-th neighbours then sit at (nearly) consecutive positions and
// a wave's gather touches ~14 cache lines instead of 27-45 on the C3D10 plate (tools/gather_lines.py)
// (84-92 registers = 5 waves per SIMD.  Forcing the budget of 6 / 7 / 8 waves -- amdgpu_waves_per_eu: 80 / 64 / 62
// registers, no spills -- is slower on every mesh: 3 GB C3D10 product 539 -> 551 / 558 / 567 us, k = 6 66 -> 72 / 79 / 86,
// profiles/r06_spmv_occupancy.txt)
template <int DM, int WPS, bool NT>
__global__ void __launch_bounds__(BS) k_spmv(int32_t nn, XcdRanges xr, const int32_t* __restrict__ slice_len,
                                             const int64_t* __restrict__ slice_off,
                                             const int32_t* __restrict__ bcol, const int32_t* __restrict__ node_of,
                                             const double* __restrict__ vals,
                                             const double* __restrict__ x, double* __restrict__ y,
                                             double* __restrict__ partials, const int32_t* __restrict__ done,
                                             const int32_t* __restrict__ slice_list, int32_t keep_permille,
                                             int32_t nreal, int32_t rot, const int32_t* __restrict__ perm,
                                             int32_t perm_rounds) {
    __shared__ double sm[BS / 64];
    __shared__ double red[(WPS > 1) ? (BS / 64) * 64 * DM : 1];
    if (done && *done) return;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)std::min<size_t>((size_t)nn * DM * sizeof(double), 0x7fffffffu), 0x00020000);
    // XCD-aware mapping: physical block b runs on XCD b % 8 (observed dispatch order; speed only).  XCD k
    // walks the contiguous slice range [xr.start[k], xr.start[k+1]), ranges balanced by stored work, so the
    // x-gathers of neighbouring slices share one private L2.  gridDim.x = 8 * blocks-per-XCD; blocks past
    // the end of their range only contribute a zero partial.
    constexpr int SPB = (BS / 64) / WPS;             // slices per workgroup task
    const int k = blockIdx.x % NXCD;
    const int bpx = gridDim.x / NXCD;                // workgroups per XCD (<= 256: the partial count stays bounded
                                                     // for any problem size; larger ranges are walked in a loop)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int part = wave % WPS;
    const int s_end = xr.start[k + 1];
    const int ntask = (s_end - xr.start[k] + SPB - 1) / SPB;
    double dot = 0.0;
    // rows are sorted by length inside windows of 64 slices: with task = b + i bpx and bpx a multiple of that period (256,
    // 512) a workgroup takes the SAME position of the window in every round -- the head of the window (C3D10: 65 blocks
    // per row against a mean of 29) always on the same workgroups, which finish long after the rest.  Where the lengths
    // spread (host: spmv_split) the rounds are rotated against each other (spmv_rot); a round still covers bpx
    // consecutive tasks, so the gathers of concurrent workgroups keep sharing the XCD's L2
    const int wgx = blockIdx.x / NXCD;
    for (int rnd = 0; rnd * bpx < ntask; ++rnd) {                     // uniform trip count within a workgroup
        const int task = rnd * bpx + (perm ? perm[((size_t)k * perm_rounds + rnd) * bpx + wgx] : spmv_rot<SLICE / SPB>(wgx, rnd, bpx, rot));
        const int spos = xr.start[k] + task * SPB + wave / WPS;
        const bool active = spos < s_end;
        const int s = (active && slice_list) ? slice_list[spos] : spos;
        double acc[DM];
#pragma unroll
        for (int r = 0; r < DM; ++r) acc[r] = 0.0;
        if (active) {
            const int32_t L = slice_len[s];
            const int32_t chunk = (L + WPS - 1) / WPS;
            const int32_t j0 = part * chunk, j1 = min(L, j0 + chunk);
            const int64_t off = slice_off[s];
            const int32_t* __restrict__ bc = bcol + off * SLICE + lane;
            constexpr int DD = DM * DM, NP = DD / 2;
            // pairs of entries as double2 (16 B per lane, 1 KiB per wave instruction); see kv_index()
            const double2* __restrict__ vp = reinterpret_cast<const double2*>(vals + off * (int64_t)(DD * SLICE)) + lane;
            const double* __restrict__ vs = vals + off * (int64_t)(DD * SLICE) + NP * (2 * SLICE) + lane;
#ifndef FEMCY_SPMV_UNROLL
#define FEMCY_SPMV_UNROLL 2
#endif
            // NT (matrix beyond the Infinity Cache): the first keep_permille/1000 of every XCD's slice range is still
            // loaded with the default policy, so that this part stays in the Infinity Cache from one product to the
            // next while the rest streams past it without allocating
            auto rows = [&](auto nt_tag) {
                constexpr bool N = decltype(nt_tag)::value;
#pragma unroll FEMCY_SPMV_UNROLL
                for (int32_t j = j0; j < j1; ++j) {
                    const int32_t col = N ? __builtin_nontemporal_load(&bc[(int64_t)j * SLICE]) : bc[(int64_t)j * SLICE];
                    double xv[DM];
                    gather_x<DM>(xrsrc, col, xv);
                    double e[DD];
#pragma unroll
                    for (int kp = 0; kp < NP; ++kp) {
                        typedef double nt_d2 __attribute__((ext_vector_type(2)));
                        const nt_d2* tp = reinterpret_cast<const nt_d2*>(&vp[(int64_t)j * (DD * SLICE / 2) + kp * SLICE]);
                        const nt_d2 t = N ? __builtin_nontemporal_load(tp) : *tp;
                        e[2 * kp] = t.x;
                        e[2 * kp + 1] = t.y;
                    }
                    if (DD & 1) e[DD - 1] = N ? __builtin_nontemporal_load(&vs[(int64_t)j * (DD * SLICE)]) : vs[(int64_t)j * (DD * SLICE)];
#pragma unroll
                    for (int r = 0; r < DM; ++r)
#pragma unroll
                        for (int cc = 0; cc < DM; ++cc) acc[r] += e[r * DM + cc] * xv[cc];
                }
            };
            if (NT && (int64_t)(spos - xr.start[k]) * 1000 >= (int64_t)keep_permille * (s_end - xr.start[k]))
                rows(std::true_type{});
            else
                rows(std::false_type{});
        }
        if (WPS > 1) {
            __syncthreads();                         // previous task's readers are done with `red`
#pragma unroll
            for (int r = 0; r < DM; ++r) red[(wave * DM + r) * 64 + lane] = acc[r];
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int w = 1; w < WPS; ++w)
#pragma unroll
                    for (int r = 0; r < DM; ++r) acc[r] += red[((wave + w) * DM + r) * 64 + lane];
            }
        }
        if (active && part == 0) {
            // node order: row permutation (SELL-C-sigma), -1 = padding lane.  Storage order (node_of == nullptr): the
            // lane's own position; padding lanes multiply zero blocks and write the zeros they are meant to hold
            const int64_t a = node_of ? (int64_t)node_of[(int64_t)s * SLICE + lane] : (int64_t)s * SLICE + lane;
            if (a >= 0) {
                const bool real = a < nreal;         // storage order: positions >= the node count are padding lanes -> 0
#pragma unroll
                for (int r = 0; r < DM; ++r) {
                    y[a * DM + r] = real ? acc[r] : 0.0;
                    if (real) dot += x[a * DM + r] * acc[r];
                }
            }
        }
    }
    if (partials) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = t;
    }
}

// ---- footprint product (round 4, storage order only; ensure_footprint in pattern.cpp).  Same partition of the work as
// k_spmv -- XCD-contiguous slice ranges, WPS waves per slice, matrix stream policy -- but x is not gathered from global
// memory block by block: the wave first stages the x entries of its FOOTPRINT (the sorted distinct positions its block
// rows refer to) in its own LDS region with coalesced 16 + 8 byte loads, then reads them from there through 16-bit
// local column indices.  A wave-level LDS sync suffices (no workgroup barrier: the region is the wave's own).
__device__ __forceinline__ void fp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int DM, int WPS, bool NT>
__global__ void __launch_bounds__(BS) k_spmv_fp(int32_t npos, XcdRanges xr, const int32_t* __restrict__ slice_len,
                                                const int64_t* __restrict__ slice_off,
                                                const uint16_t* __restrict__ lcol, const int32_t* __restrict__ fp_ptr,
                                                const int32_t* __restrict__ fp, const double* __restrict__ vals,
                                                const double* __restrict__ x, double* __restrict__ y,
                                                double* __restrict__ partials, const int32_t* __restrict__ done,
                                                int32_t keep_permille, int32_t nreal, int32_t fcap, int32_t rot,
                                                const int32_t* __restrict__ perm, int32_t perm_rounds) {
    extern __shared__ __attribute__((aligned(16))) double xs_all[];      // [BS / 64][fcap][DM]
    __shared__ double sm[BS / 64];
    __shared__ double red[(WPS > 1) ? (BS / 64) * 64 * DM : 1];
    if (done && *done) return;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)std::min<size_t>((size_t)npos * DM * sizeof(double), 0x7fffffffu), 0x00020000);
    constexpr int SPB = (BS / 64) / WPS;
    constexpr int DD = DM * DM, NP = DD / 2;
    const int k = blockIdx.x % NXCD;
    const int bpx = gridDim.x / NXCD;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int part = wave % WPS;
    double* __restrict__ xs = xs_all + (size_t)wave * fcap * DM;
    const int s_end = xr.start[k + 1];
    const int ntask = (s_end - xr.start[k] + SPB - 1) / SPB;
    double dot = 0.0;
    const int wgx = blockIdx.x / NXCD;
    for (int rnd = 0; rnd * bpx < ntask; ++rnd) {                     // rotated rounds: see k_spmv
        const int task = rnd * bpx + (perm ? perm[((size_t)k * perm_rounds + rnd) * bpx + wgx] : spmv_rot<SLICE / SPB>(wgx, rnd, bpx, rot));
        const int s = xr.start[k] + task * SPB + wave / WPS;
        const bool active = s < s_end;
        double acc[DM];
#pragma unroll
        for (int r = 0; r < DM; ++r) acc[r] = 0.0;
        if (active) {
            const int32_t L = slice_len[s];
            const int32_t chunk = (L + WPS - 1) / WPS;
            const int32_t j0 = part * chunk, j1 = min(L, j0 + chunk);
            const int64_t off = slice_off[s];
            // ---- stage the footprint (the previous task's reads of this region are complete: same wave, program order)
            const int32_t f0 = fp_ptr[(int64_t)s * WPS + part], F = fp_ptr[(int64_t)s * WPS + part + 1] - f0;
            fp_wave_sync();
            for (int32_t i = lane; i < F; i += 64) {
                const int32_t p = NT ? __builtin_nontemporal_load(&fp[f0 + i]) : fp[f0 + i];
                double xv[DM];
                gather_x<DM>(xrsrc, p, xv);
#pragma unroll
                for (int cc = 0; cc < DM; ++cc) xs[i * DM + cc] = xv[cc];
            }
            fp_wave_sync();
            const uint16_t* __restrict__ lc = lcol + off * SLICE + lane;
            const double2* __restrict__ vp = reinterpret_cast<const double2*>(vals + off * (int64_t)(DD * SLICE)) + lane;
            const double* __restrict__ vs = vals + off * (int64_t)(DD * SLICE) + NP * (2 * SLICE) + lane;
            auto rows = [&](auto nt_tag) {
                constexpr bool N = decltype(nt_tag)::value;
#pragma unroll FEMCY_SPMV_UNROLL
                for (int32_t j = j0; j < j1; ++j) {
                    const int32_t col = N ? __builtin_nontemporal_load(&lc[(int64_t)j * SLICE]) : lc[(int64_t)j * SLICE];
                    double e[DD];
#pragma unroll
                    for (int kp = 0; kp < NP; ++kp) {
                        typedef double nt_d2 __attribute__((ext_vector_type(2)));
                        const nt_d2* tp = reinterpret_cast<const nt_d2*>(&vp[(int64_t)j * (DD * SLICE / 2) + kp * SLICE]);
                        const nt_d2 t = N ? __builtin_nontemporal_load(tp) : *tp;
                        e[2 * kp] = t.x;
                        e[2 * kp + 1] = t.y;
                    }
                    if (DD & 1) e[DD - 1] = N ? __builtin_nontemporal_load(&vs[(int64_t)j * (DD * SLICE)]) : vs[(int64_t)j * (DD * SLICE)];
                    double xv[DM];
#pragma unroll
                    for (int cc = 0; cc < DM; ++cc) xv[cc] = xs[col * DM + cc];
#pragma unroll
                    for (int r = 0; r < DM; ++r)
#pragma unroll
                        for (int cc = 0; cc < DM; ++cc) acc[r] += e[r * DM + cc] * xv[cc];
                }
            };
            if (NT && (int64_t)(s - xr.start[k]) * 1000 >= (int64_t)keep_permille * (s_end - xr.start[k]))
                rows(std::true_type{});
            else
                rows(std::false_type{});
        }
        if (WPS > 1) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < DM; ++r) red[(wave * DM + r) * 64 + lane] = acc[r];
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int w = 1; w < WPS; ++w)
#pragma unroll
                    for (int r = 0; r < DM; ++r) acc[r] += red[((wave + w) * DM + r) * 64 + lane];
            }
        }
        if (active && part == 0) {
            const int64_t a = (int64_t)s * SLICE + lane;
            const bool real = a < nreal;
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                y[a * DM + r] = real ? acc[r] : 0.0;
                if (real) dot += x[a * DM + r] * acc[r];
            }
        }
    }
    if (partials) {
        const double t = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = t;
    }
}

// M = 1/diag(K) (M_init, conjugateGradientSolver.py:48-51): diagonal block is stored row 0 of the slice
template <int DM>
__global__ void __launch_bounds__(BS) k_jacobi(int32_t nn, const int32_t* __restrict__ pos,
                                               const int64_t* __restrict__ slice_off,
                                               const double* __restrict__ vals, double* __restrict__ M, int invert) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nn) return;
    const int32_t pa = pos[a];
    const int64_t row = slice_off[pa >> 6];
    const int lane = pa & 63;
#pragma unroll
    for (int r = 0; r < DM; ++r) {
        const double dg = vals[kv_index<DM>(row, r * DM + r, lane)];
        M[a * DM + r] = invert ? 1.0 / dg : dg;
    }
}
// the same in storage order: entry p belongs to position p (coalesced); padding lanes get M = 0, so r = d = 0 there
template <int DM>
__global__ void __launch_bounds__(BS) k_jacobi_pos(int32_t npos, const int32_t* __restrict__ node_of,
                                                   const int64_t* __restrict__ slice_off,
                                                   const double* __restrict__ vals, double* __restrict__ M) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const bool real = node_of[p] >= 0;
    const int64_t row = slice_off[p >> 6];
    const int lane = (int)(p & 63);
#pragma unroll
    for (int r = 0; r < DM; ++r) M[p * DM + r] = real ? 1.0 / vals[kv_index<DM>(row, r * DM + r, lane)] : 0.0;
}
// node order <-> storage order of a vector (once per solve each way)
template <int DM>
__global__ void __launch_bounds__(BS) k_to_pos(int32_t npos, const int32_t* __restrict__ node_of,
                                               const double* __restrict__ v, double* __restrict__ vp) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int64_t a = node_of[p];
#pragma unroll
    for (int r = 0; r < DM; ++r) vp[p * DM + r] = a >= 0 ? v[a * DM + r] : 0.0;
}
template <int DM>
__global__ void __launch_bounds__(BS) k_from_pos(int32_t npos, const int32_t* __restrict__ node_of,
                                                 const double* __restrict__ vp, double* __restrict__ v) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npos) return;
    const int64_t a = node_of[p];
    if (a < 0) return;
#pragma unroll
    for (int r = 0; r < DM; ++r) v[a * DM + r] = vp[p * DM + r];
}
__global__ void __launch_bounds__(BS) k_recip(int64_t n, double* __restrict__ v) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) v[i] = 1.0 / v[i];
}

// x = 0, r = b, d = M r, partials of (r.M.r, max|r|)   (re_init + r_d_init, :32-38, :60-65)
__global__ void __launch_bounds__(BS) k_pcg_init(int64_t n2, const double2* __restrict__ b, const double2* __restrict__ M,
                                                 double2* __restrict__ x, double2* __restrict__ r,
                                                 double2* __restrict__ d, const uint8_t* __restrict__ owner,
                                                 double* __restrict__ part2) {
    __shared__ double sm1[BS / 64], sm2[BS / 64];
    double rMr = 0.0, rm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += (int64_t)gridDim.x * BS) {
        const double2 bv = b[i], mv = M[i];
        x[i] = make_double2(0.0, 0.0);
        r[i] = bv;
        d[i] = make_double2(mv.x * bv.x, mv.y * bv.y);
        double w0 = 1.0, w1 = 1.0;
        if (owner) {
            w0 = owner[2 * i];
            w1 = owner[2 * i + 1];
        }
        rMr += w0 * (bv.x * mv.x * bv.x) + w1 * (bv.y * mv.y * bv.y);
        rm = fmax(rm, fmax(nan_to_inf_abs(bv.x), nan_to_inf_abs(bv.y)));
    }
    const double s = block_sum(rMr, sm1), m = block_max(rm, sm2);
    if (threadIdx.x == 0) {
        part2[2 * blockIdx.x] = s;
        part2[2 * blockIdx.x + 1] = m;
    }
}

__global__ void __launch_bounds__(BS) k_pcg_init_final(int np, const double* __restrict__ part2, PcgState* st,
                                                       const double* __restrict__ gathered, int nranks, double eps) {
    __shared__ double sm1[BS / 64], sm2[BS / 64];
    double s = 0.0, m = 0.0;
    if (gathered) {   // multi-rank: (sum, max) pairs already reduced per rank
        for (int i = threadIdx.x; i < nranks; i += BS) {
            s += gathered[2 * i];
            m = fmax(m, gathered[2 * i + 1]);
        }
    } else {
        for (int i = threadIdx.x; i < np; i += BS) {
            s += part2[2 * i];
            m = fmax(m, part2[2 * i + 1]);
        }
    }
    s = block_sum(s, sm1);
    m = block_max(m, sm2);
    if (threadIdx.x == 0) {
        st->rMr[0] = s;
        st->rMr[1] = 0.0;
        st->r0 = m;
        st->rmax = m;
        st->dAd = 0.0;
        st->iters = 0;
        st->it_k3 = 0;
        st->eps = eps;
        st->done = (m == 0.0) ? 1 : ((m != m || isinf(m)) ? 2 : 0);
        st->skip = st->done ? 1 : 0;
    }
}

// x += alpha d; r -= alpha Ad; partials (r.M.r, max|r|)
// streaming accesses of the PCG vector kernels; NT = non-temporal (keeps the Infinity-Cache-resident matrix from being
// displaced by vectors that are re-read at most once per iteration)
typedef double femcy_d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld2(const double2* p) {
    if (NT) {
        const femcy_d2v t = __builtin_nontemporal_load(reinterpret_cast<const femcy_d2v*>(p));
        return make_double2(t.x, t.y);
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st2(double2* p, double2 v) {
    if (NT) {
        femcy_d2v t;
        t.x = v.x;
        t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<femcy_d2v*>(p));
    } else {
        *p = v;
    }
}

// Element-wise PCG kernels.  A thread owns up to VU double2 elements per batch, base + u * stride, and issues the
// loads of the whole batch before anything else, so that one memory round trip (instead of one per element) hides
// under the scalar prologue that re-reduces the producer's partials.  At 548 535 DOF and the default 512-workgroup
// cap a thread holds 2.09 elements: one batch.
constexpr int VU = 4;
constexpr int PU = 4;    // d.Ad partials prefetched per thread (covers 1024 SpMV workgroups; more are looped)
constexpr int PU2 = 2;   // (r.M.r, max|r|) pairs prefetched per thread (covers the default 512-workgroup cap)

// r -= alpha Ad with alpha = r.M.r / d.Ad, partials of (r.M.r, max|r|) of the new r; alpha is published for k_update_d,
// which applies x += alpha d in the pass where it reads d anyway (x would otherwise be the only reason for this
// kernel to load d: 8 n bytes per iteration).
// Stop flag: `done` is written by k_update_d only and tested by k_spmv / k_update_xr only; k_update_xr hands it to
// k_update_d as `skip`.  No kernel tests a word that a workgroup of the same launch writes (a late workgroup of
// k_update_d could otherwise see the flag its block 0 had just set and drop the x / d update of the converging
// iteration).
template <bool NT, bool MULTI>
__global__ void __launch_bounds__(BS) k_update_xr(XcdRanges er, int np1, const double* __restrict__ part1,
                                                  const double* __restrict__ dAd_reduced, PcgState* st,
                                                  const double2* __restrict__ Ad, const double2* __restrict__ M,
                                                  double2* __restrict__ r, const uint8_t* __restrict__ owner,
                                                  double* __restrict__ part2) {
    __shared__ double sm1[BS / 64], sm2[BS / 64];
    if (st->done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st->skip = 1;
        return;
    }
    // XCD-aligned element ranges (er, in double2 units): workgroup b runs on XCD b % 8 (observed; speed only) and
    // works on the entries whose matrix rows that XCD multiplies in k_spmv -- Ad (written there), r and d (gathered
    // there by the next product) then stay in that XCD's L2 from kernel to kernel instead of crossing the fabric
    const int xk = blockIdx.x % NXCD;
    const int64_t stride = (int64_t)(gridDim.x / NXCD) * BS;
    const int64_t n2 = er.start[xk + 1];
    int64_t base = (int64_t)er.start[xk] + (int64_t)(blockIdx.x / NXCD) * BS + threadIdx.x;
    // the producer's partials first (VMEM returns in order: loaded after the batch they would only arrive behind it)
    double pv[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int k = threadIdx.x + u * BS;            // branch-free (out-of-range lanes read the always-zero slot
        pv[u] = MULTI ? 0.0 : part1[k < np1 ? k : MAX_PARTIALS];   // behind the array): the compiler can then count the
    }                                                  // loads -- vmcnt(N) instead of a full drain before the first use
    double2 av[VU], mv[VU], rv[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
        const int64_t i = max((int64_t)er.start[xk], min(base + u * stride, n2 - 1));
        av[u] = ld2<NT>(Ad + i);
        mv[u] = ld2<NT>(M + i);
        rv[u] = ld2<NT>(r + i);
    }
    const int it = st->iters;                       // stable: written by the previous iteration's k_update_d
    double dAd;
    if (MULTI) {                                    // multi-rank: d.Ad was reduced over the ranks by the exchange
        dAd = *dAd_reduced;
    } else {
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < PU; ++u) v += pv[u];
        for (int k = threadIdx.x + PU * BS; k < np1; k += BS) v += part1[k];
        dAd = block_sum(v, sm1);
    }
    const double alpha = st->rMr[it & 1] / dAd;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->it_k3 = it;
        st->alpha = alpha;
        st->skip = 0;
    }
    double rMr = 0.0, rm = 0.0;
    while (base < n2) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int64_t i = base + u * stride;
            if (i < n2) {
                rv[u].x = rv[u].x - alpha * av[u].x;
                rv[u].y = rv[u].y - alpha * av[u].y;
                st2<NT>(r + i, rv[u]);
                double w0 = 1.0, w1 = 1.0;
                if (owner) {
                    w0 = owner[2 * i];
                    w1 = owner[2 * i + 1];
                }
                rMr += w0 * (rv[u].x * mv[u].x * rv[u].x) + w1 * (rv[u].y * mv[u].y * rv[u].y);
                rm = fmax(rm, fmax(nan_to_inf_abs(rv[u].x), nan_to_inf_abs(rv[u].y)));
            }
        }
        base += VU * stride;
        if (base < n2) {
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int64_t i = base + u * stride;
                if (i < n2) {
                    av[u] = ld2<NT>(Ad + i);
                    mv[u] = ld2<NT>(M + i);
                    rv[u] = ld2<NT>(r + i);
                }
            }
        }
    }
    const double s = block_sum(rMr, sm1), m = block_max(rm, sm2);
    if (threadIdx.x == 0) {
        part2[2 * blockIdx.x] = s;
        part2[2 * blockIdx.x + 1] = m;
    }
    // (round 4: letting the last-arriving workgroup reduce the pairs of a multi-rank run -- write-through partials, a
    // ticket per workgroup -- instead of the single-block launch behind this kernel was built and measured: the fan-in
    // of 512 tickets costs more than the launch it saves, 53.2 against 49.8 us per iteration; dropped)
}

// x += alpha d (this iteration's alpha, old d), then d = M r + beta d; publish scalars and the stopping decision
template <bool NT>
__global__ void __launch_bounds__(BS) k_update_d(XcdRanges er, int np2, const double* __restrict__ part2,
                                                 const double* __restrict__ gathered, int nranks, PcgState* st, const double2* __restrict__ r,
                                                 const double2* __restrict__ M, double2* __restrict__ d,
                                                 double2* __restrict__ x) {
    __shared__ double sm1[BS / 64], sm2[BS / 64];
    if (st->skip) return;                           // written by this iteration's k_update_xr, never by this kernel
    const int xk = blockIdx.x % NXCD;               // XCD-aligned element ranges: see k_update_xr
    const int64_t stride = (int64_t)(gridDim.x / NXCD) * BS;
    const int64_t n2 = er.start[xk + 1];
    int64_t base = (int64_t)er.start[xk] + (int64_t)(blockIdx.x / NXCD) * BS + threadIdx.x;
    // (r.M.r, max|r|) pairs of k_update_xr first, then the batch (in-order VMEM returns, see k_update_xr)
    const double* __restrict__ pairs = gathered ? gathered : part2;
    const int npairs = gathered ? nranks : np2;
    const int zslot = gathered ? nranks : MAX_PARTIALS;   // a (0, 0) pair behind either array that nothing ever writes
    double2 pp[PU2];
#pragma unroll
    for (int u = 0; u < PU2; ++u) {
        const int k = threadIdx.x + u * BS;            // branch-free through the zero slot (see k_update_xr)
        pp[u] = *reinterpret_cast<const double2*>(pairs + 2 * (k < npairs ? k : zslot));
    }
    double2 rv[VU], mv[VU], dv[VU], xv[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {                  // first batch in flight while the scalars are reduced
        const int64_t i = max((int64_t)er.start[xk], min(base + u * stride, n2 - 1));
        rv[u] = ld2<NT>(r + i);
        mv[u] = ld2<NT>(M + i);
        dv[u] = ld2<NT>(d + i);
        xv[u] = ld2<NT>(x + i);
    }
    double s = 0.0, m = 0.0;
#pragma unroll
    for (int u = 0; u < PU2; ++u) {
        s += pp[u].x;
        m = fmax(m, pp[u].y);
    }
    for (int k = threadIdx.x + PU2 * BS; k < npairs; k += BS) {
        s += pairs[2 * k];
        m = fmax(m, pairs[2 * k + 1]);
    }
    const double rMr_new = block_sum(s, sm1);
    const double rmax = block_max(m, sm2);
    const int it = st->it_k3;                       // stable: written by this iteration's k_update_xr
    const double alpha = st->alpha;                 // likewise
    const double eps = st->eps;
    const double rMr_old = st->rMr[it & 1];
    const double r0 = st->r0;
    const double beta = rMr_new / rMr_old;
    while (base < n2) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int64_t i = base + u * stride;
            if (i < n2) {
                xv[u].x = xv[u].x + alpha * dv[u].x;
                xv[u].y = xv[u].y + alpha * dv[u].y;
                dv[u].x = mv[u].x * rv[u].x + beta * dv[u].x;
                dv[u].y = mv[u].y * rv[u].y + beta * dv[u].y;
                st2<NT>(x + i, xv[u]);
                st2<NT>(d + i, dv[u]);
            }
        }
        base += VU * stride;
        if (base < n2) {
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int64_t i = base + u * stride;
                if (i < n2) {
                    rv[u] = ld2<NT>(r + i);
                    mv[u] = ld2<NT>(M + i);
                    dv[u] = ld2<NT>(d + i);
                    xv[u] = ld2<NT>(x + i);
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rMr[(it + 1) & 1] = rMr_new;
        st->rmax = rmax;
        st->iters = it + 1;
        if (rmax != rmax || isinf(rmax) || rMr_new != rMr_new)
            st->done = 2;
        else if (rmax < eps * r0)
            st->done = 1;
    }
}

// ---- round 4: the two vector kernels of an iteration as ONE launch (single rank; FEMCY_OPT_PCG_FUSED_UPDATE, default
// OFF: measured 1.5 us per iteration SLOWER than the two kernels on MI355X -- a kernel boundary costs ~1.5 us here, a
// grid-wide exchange 2.6-3.3 us, and the loads the boundary lets the second kernel issue early are serialised behind
// the exchange; profiles/r04_ab_fused_update.txt.  Kept as a tested option and as the record of that measurement).  k_update_xr and k_update_d are
// latency-bound launches (5.6-6.7 us each for 18-26 MB) separated by a kernel boundary whose only purpose is the
// grid-wide (r.M.r, max|r|); here that reduction is an in-kernel exchange of tagged granules (granule.hpp: one 16-byte
// write-through store per workgroup, one wave sweeps), r and M stay in registers across it and d / x are already
// in flight when it starts: one boundary and 16 n bytes less per iteration.  Every thread owns <= U double2 of its
// XCD's element range.  The tag is a launch counter kept in PcgState (never reset: tags grow for the life of the
// context), so no granule is ever re-armed; every workgroup of the launch is resident (checked by the host) and the
// sweep is bounded -- a time-out poisons the granules, sets done = 3 and the host redoes the solve with the two
// kernels and stays there.
template <bool NT, int U>
__global__ void __launch_bounds__(BS) k_update_fused(XcdRanges er, int np1, const double* __restrict__ part1, PcgState* st,
                                                     const double2* __restrict__ Ad, const double2* __restrict__ M,
                                                     double2* __restrict__ r, double2* __restrict__ d,
                                                     double2* __restrict__ x, double* __restrict__ slots,
                                                     uint32_t spin_limit) {
    __shared__ double sm1[BS / 64], sm2[BS / 64], bc[2];
    __shared__ int s_fail;
    if (st->done) return;          // written by an earlier launch only: every workgroup of this one reads the same value
    const int G = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xk = blockIdx.x % NXCD;
    const int64_t stride = (int64_t)(G / NXCD) * BS;
    const int64_t lo = er.start[xk], hi = er.start[xk + 1];
    const int64_t base = lo + (int64_t)(blockIdx.x / NXCD) * BS + tid;
    if (tid == 0) s_fail = 0;
    double pv[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int k = tid + u * BS;
        pv[u] = part1[k < np1 ? k : MAX_PARTIALS];
    }
    double2 av[U], mv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = max(lo, min(base + u * stride, hi - 1));
        av[u] = ld2<NT>(Ad + i);
        mv[u] = ld2<NT>(M + i);
        rv[u] = ld2<NT>(r + i);
    }
    const int it = st->iters;
    const unsigned long long tag = st->xround + 1;      // stable: written by block 0 of the previous launch at its end
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < PU; ++u) v += pv[u];
    for (int k = tid + PU * BS; k < np1; k += BS) v += part1[k];
    const double dAd = block_sum(v, sm1);
    const double rMr_old = st->rMr[it & 1];
    const double alpha = rMr_old / dAd;
    double rMr = 0.0, rm = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + u * stride;
        if (i < hi) {
            rv[u].x = rv[u].x - alpha * av[u].x;
            rv[u].y = rv[u].y - alpha * av[u].y;
            st2<NT>(r + i, rv[u]);
            rMr += rv[u].x * mv[u].x * rv[u].x + rv[u].y * mv[u].y * rv[u].y;
            rm = fmax(rm, fmax(nan_to_inf_abs(rv[u].x), nan_to_inf_abs(rv[u].y)));
        }
    }
    // d and x of this thread: requested before the exchange, used behind it (av's registers are free now)
    double2 dv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = max(lo, min(base + u * stride, hi - 1));
        dv[u] = ld2<NT>(d + i);
        xv[u] = ld2<NT>(x + i);
    }
    const double s = block_sum(rMr, sm1), m = block_max(rm, sm2);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, G * 2 * 16, 0x00020000);
    if (wave == 0) {
        if (lane == 0) granule_store(srsrc, 2 * blockIdx.x, s, tag);
        if (lane == 1) granule_store(srsrc, 2 * blockIdx.x + 1, m, tag);
        double o[2];
        const int op[2] = {0, 1};
        const bool okx = granule_sweep<2>(srsrc, 0, G, tag, spin_limit, o, op);
        if (!okx && lane < 2) granule_store(srsrc, 2 * blockIdx.x + lane, 0.0, TAG_POISON);
        if (lane == 0) {
            bc[0] = o[0];
            bc[1] = o[1];
            if (!okx) s_fail = 1;
        }
    }
    __syncthreads();
    if (s_fail) {
        // the time-out verdict wins over whatever workgroup 0 concludes from a sweep that happened to complete for it
        // (the granule reads of one sweep are not atomic: verdicts can be mixed inside one launch)
        if (tid == 0) atomicMax(&st->done, 3);
        return;
    }
    const double rMr_new = bc[0], rmax = bc[1];
    const double beta = rMr_new / rMr_old;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + u * stride;
        if (i < hi) {
            xv[u].x = xv[u].x + alpha * dv[u].x;
            xv[u].y = xv[u].y + alpha * dv[u].y;
            dv[u].x = mv[u].x * rv[u].x + beta * dv[u].x;
            dv[u].y = mv[u].y * rv[u].y + beta * dv[u].y;
            st2<NT>(x + i, xv[u]);
            st2<NT>(d + i, dv[u]);
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        st->rMr[(it + 1) & 1] = rMr_new;
        st->rmax = rmax;
        st->alpha = alpha;
        st->iters = it + 1;
        st->xround = tag;
        if (rmax != rmax || isinf(rmax) || rMr_new != rMr_new)
            atomicMax(&st->done, 2);
        else if (rmax < st->eps * st->r0)
            atomicMax(&st->done, 1);
    }
}

// single-block reduction of the (r.M.r, max|r|) partials into one pair (multi-rank path)
__global__ void __launch_bounds__(BS) k_sum_partials2(int np, const double* __restrict__ part2, double* out2) {
    __shared__ double sm1[BS / 64], sm2[BS / 64];
    double s = 0.0, m = 0.0;
    for (int i = threadIdx.x; i < np; i += BS) {
        s += part2[2 * i];
        m = fmax(m, part2[2 * i + 1]);
    }
    s = block_sum(s, sm1);
    m = block_max(m, sm2);
    if (threadIdx.x == 0) {
        out2[0] = s;
        out2[1] = m;
    }
}
// packed global interface vector in one launch: buf[s] = v[local dof of slot s] (0 where this rank does not hold
// the slot -- the all-reduce sums the contributions of the sharing ranks); the extra last workgroup reduces the
// SpMV partials into buf[nslots] (local d.K_loc.d), so one collective carries both.
__global__ void __launch_bounds__(BS) k_iface_pack_all(int32_t nslots, const int32_t* __restrict__ slot2dof,
                                                       const double* __restrict__ v, double* __restrict__ buf, int np,
                                                       const double* __restrict__ part) {
    __shared__ double sm[BS / 64];
    if ((int)blockIdx.x == (int)gridDim.x - 1) {
        if (part) {
            const double t = reduce_partials_sum(part, np, sm);
            if (threadIdx.x == 0) buf[nslots] = t;
        }
        return;
    }
    const int32_t s = blockIdx.x * BS + threadIdx.x;
    if (s < nslots) {
        const int32_t d = slot2dof[s];
        buf[s] = d >= 0 ? v[d] : 0.0;
    }
}
__global__ void k_iface_unpack(int32_t k, const int32_t* __restrict__ dof, const int32_t* __restrict__ slot,
                               const double* __restrict__ buf, double* __restrict__ v) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) v[dof[i]] = buf[slot[i]];
}

// neighbour exchange: send[j] = v[nb_dofs[j]]; after the exchange every local interface DOF sums its own value and
// the received ones in ascending RANK order -- all ranks holding the DOF add the same numbers in the same order, so
// the replicas stay bit-identical
__global__ void k_p2p_pack(int32_t total, const int32_t* __restrict__ nb_dofs, const double* __restrict__ v,
                           double* __restrict__ send) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < total) send[j] = v[nb_dofs[j]];
}
__global__ void k_p2p_sum(int32_t k, const int32_t* __restrict__ dof, const int32_t* __restrict__ ptr,
                          const int32_t* __restrict__ src, const double* __restrict__ recv, double* __restrict__ v) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int32_t d = dof[i];
    const double own = v[d];
    double s = 0.0;
    for (int32_t t = ptr[i]; t < ptr[i + 1]; ++t) {
        const int32_t q = src[t];
        s += q < 0 ? own : recv[q];
    }
    v[d] = s;
}

// ------------------------------------------------------------------------------- vector helpers
__global__ void __launch_bounds__(BS) k_fill(int64_t n, double* __restrict__ v, double val) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) v[i] = val;
}
__global__ void __launch_bounds__(BS) k_sub(int64_t n, double* __restrict__ c, const double* __restrict__ a,
                                            const double* __restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) c[i] = a[i] - b[i];
}
__global__ void __launch_bounds__(BS) k_axpy(int64_t n, double* __restrict__ a, const double* __restrict__ b, double cc,
                                             const double* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS)
        a[i] = b[i] + cc * d[i];
}
__global__ void __launch_bounds__(BS) k_scale(int64_t n, double* __restrict__ v, double s) {
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) v[i] *= s;
}
__global__ void k_scatter(int32_t k, const int32_t* __restrict__ idx, const double* __restrict__ vals,
                          double* __restrict__ v) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) v[idx[i]] = vals[i];
}
__global__ void k_scatter_const(int32_t k, const int32_t* __restrict__ idx, double val, double* __restrict__ v) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) v[idx[i]] = val;
}
// mode 0: sum of squares, mode 1: max |.|; partials then a single-block finish into *out
__global__ void __launch_bounds__(BS) k_reduce(int64_t n, const double* __restrict__ v, int mode,
                                               const uint8_t* __restrict__ owner, double* __restrict__ part) {
    __shared__ double sm[BS / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n; i += (int64_t)gridDim.x * BS) {
        if (owner && !owner[i]) continue;           // multi-rank: a shared DOF counts on its owner only
        const double t = v[i];
        acc = mode ? fmax(acc, nan_to_inf_abs(t)) : acc + t * t;
    }
    const double t = mode ? block_max(acc, sm) : block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void __launch_bounds__(BS) k_reduce_final(int np, const double* __restrict__ part, int mode, double* out) {
    __shared__ double sm[BS / 64];
    const double t = mode ? reduce_partials_max(part, np, sm) : reduce_partials_sum(part, np, sm);
    if (threadIdx.x == 0) *out = t;
}

// FEMCY_OPT_EW_GRID caps the workgroups of the element-wise kernels (default 512 = 2 per CU: measured 4 % faster per
// CG iteration than one element per thread, fewer partials to re-reduce)
static inline int ew_grid(const Ctx* c, int64_t n) {
    int64_t g = (n + BS - 1) / BS;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, c->ew_cap));
}

int vec_fill(Ctx* c, double* d, double v, int64_t n) {
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(c, n)), dim3(BS), 0, c->stream, n, d, v);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
int vec_sub(Ctx* c, double* dc, const double* da, const double* db) {
    hipLaunchKernelGGL(k_sub, dim3(ew_grid(c, c->n)), dim3(BS), 0, c->stream, c->n, dc, da, db);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
int vec_axpy(Ctx* c, double* da, const double* db, double cc, const double* dd) {
    hipLaunchKernelGGL(k_axpy, dim3(ew_grid(c, c->n)), dim3(BS), 0, c->stream, c->n, da, db, cc, dd);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
int vec_scale(Ctx* c, double* d, double s) {
    hipLaunchKernelGGL(k_scale, dim3(ew_grid(c, c->n)), dim3(BS), 0, c->stream, c->n, d, s);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
int vec_scatter(Ctx* c, double* d, const int32_t* d_idx, const double* d_vals, int32_t k) {
    if (k <= 0) return FEMCY_OK;
    hipLaunchKernelGGL(k_scatter, dim3((k + BS - 1) / BS), dim3(BS), 0, c->stream, k, d_idx, d_vals, d);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
int vec_scatter_const(Ctx* c, double* d, const int32_t* d_idx, double val, int32_t k) {
    if (k <= 0) return FEMCY_OK;
    hipLaunchKernelGGL(k_scatter_const, dim3((k + BS - 1) / BS), dim3(BS), 0, c->stream, k, d_idx, val, d);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}
// the scalar at d_val (device) -> host; multi-rank: summed (mode 0) or maximised (mode 1) over the ranks first.
// Collective when a communicator is attached: every rank must make the same call.
int scalar_across_ranks(Ctx* c, double* d_val, int mode, double* out) {
    if (c->comm && mode == 0) {
        int rc = comm_allreduce_sum(c, d_val, 1);
        if (rc) return rc;
    }
    if (c->comm && mode == 1) {
        int rc = comm_allgather(c, d_val, c->d_gather, 1);
        if (rc) return rc;
        FEMCY_HIP(hipMemcpyAsync(c->h_scalar, c->d_gather, sizeof(double) * c->nranks, hipMemcpyDeviceToHost, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        double m = 0.0;
        for (int r = 0; r < c->nranks; ++r) m = std::fmax(m, c->h_scalar[r]);
        *out = m;
        return FEMCY_OK;
    }
    FEMCY_HIP(hipMemcpyAsync(c->h_scalar, d_val, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    *out = c->h_scalar[0];
    return FEMCY_OK;
}

static int vec_reduce(Ctx* c, const double* d, int mode, double* out) {
    const int g = ew_grid(c, c->n);
    hipLaunchKernelGGL(k_reduce, dim3(g), dim3(BS), 0, c->stream, c->n, d, mode,
                       (const uint8_t*)(c->comm ? c->d_owner : nullptr), c->d_part1);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(BS), 0, c->stream, g, c->d_part1, mode, c->d_part2);
    FEMCY_HIP(hipGetLastError());
    return scalar_across_ranks(c, c->d_part2, mode, out);
}
int vec_sumsq(Ctx* c, const double* d, double* out) { return vec_reduce(c, d, 0, out); }
int vec_absmax(Ctx* c, const double* d, double* out) { return vec_reduce(c, d, 1, out); }

// ---------------------------------------------------------------------------------------- SpMV
__global__ void k_fence_noop() {}

static int launch_spmv_impl(Ctx* c, const double* d_x, double* d_y, double* d_partials, int* nblocks_out,
                            const XcdRanges& xr, int grid, const int32_t* slice_list, int part_off = 0,
                            bool pos_space = false) {
    // the partials of a launch go to d_partials[part_off .. part_off + grid): the two halves of a split product share
    // one array of MAX_PARTIALS entries
    if (part_off + grid > MAX_PARTIALS) {
        set_error("SpMV grid %d at partial offset %d exceeds MAX_PARTIALS %d", grid, part_off, MAX_PARTIALS);
        return FEMCY_EINVAL;
    }
    const int32_t* done = d_partials ? &c->d_state->done : nullptr;
    // the balanced task lists (spmv_split) belong to the product over all slices in c->xcd's ranges and c->spmv_grid
    const int32_t* spmv_perm = (c->spmv_rot == 64 && c->spmv_perm_rounds > 0 && !slice_list && grid == c->spmv_grid &&
                                xr.start[NXCD] == c->xcd.start[NXCD]) ? c->d_spmv_perm : nullptr;
    // timing: start/stop events attached to the dispatch itself (hipExtLaunchKernel), i.e. the kernel's own
    // begin/end timestamps -- no marker packets between the PCG kernels, agrees with rocprofv3's kernel trace
    // FEMCY_OPT_TIMING = k > 1 samples every k-th SpMV launch: a profiled dispatch costs ~5 us of pipeline
    // drain, which would otherwise inflate every CG iteration of a timed run by ~10 %
    const bool sample = !slice_list &&
                        (c->opt_timing == 1 || (c->opt_timing > 1 && (c->spmv_count++ % c->opt_timing) == 0));
    EventPair* ev = sample ? timing_acquire(c, T_SPMV) : nullptr;
    hipEvent_t ea = ev ? ev->a : nullptr, eb = ev ? ev->b : nullptr;
    // a sampled dispatch is preceded by an empty kernel: the profiled start stamp is taken when the packet is
    // picked up, i.e. possibly while the previous PCG kernel is still draining; the empty kernel absorbs that wait
    if (ev && c->opt_timing_fence) hipLaunchKernelGGL(k_fence_noop, dim3(1), dim3(64), 0, c->stream);
#define SPMV_ARGS                                                                                              \
    (pos_space ? c->nslices * SLICE : c->nn), xr, (const int32_t*)c->d_slice_len, (const int64_t*)c->d_slice_off,  \
        (const int32_t*)(pos_space ? c->d_bcolp : c->d_bcol), (const int32_t*)(pos_space ? nullptr : c->d_node_of),  \
        (const double*)c->d_Kvals, d_x, d_y, d_partials, done, slice_list, (int32_t)c->spmv_keep_permille, c->nn, (int32_t)(c->spmv_rot == 64 ? 19 : c->spmv_rot), spmv_perm, c->spmv_perm_rounds
#define SPMV_LAUNCH_NT(DM_, WPS_, NT_)                                                                         \
    do {                                                                                                       \
        if (ev)                                                                                                \
            hipExtLaunchKernelGGL((k_spmv<DM_, WPS_, NT_>), dim3(grid), dim3(BS), 0, c->stream, ea, eb, 0, SPMV_ARGS); \
        else /* plain launch: also the form that is captured into the PCG hipGraph */                          \
            hipLaunchKernelGGL((k_spmv<DM_, WPS_, NT_>), dim3(grid), dim3(BS), 0, c->stream, SPMV_ARGS);        \
    } while (0)
#define SPMV_LAUNCH(DM_, WPS_)                          \
    do {                                                \
        if (c->spmv_nt) SPMV_LAUNCH_NT(DM_, WPS_, true); \
        else SPMV_LAUNCH_NT(DM_, WPS_, false);          \
    } while (0)
    const int wps = c->spmv_wps;
    // storage order, single launch over all slices: the footprint product where its arrays exist and fit the LDS
    if (pos_space && !slice_list && c->opt_spmv_fp) {
        int rc = ensure_footprint(c);
        if (rc) return rc;
        const size_t lds = (size_t)(BS / 64) * c->fp_cap * c->dm * sizeof(double);
        if (c->fp_cap > 0 && lds <= (size_t)60 * 1024) {
#define SPMV_FP_ARGS                                                                                            \
    c->nslices * SLICE, xr, (const int32_t*)c->d_slice_len, (const int64_t*)c->d_slice_off, (const uint16_t*)c->d_lcol,  \
        (const int32_t*)c->d_fp_ptr, (const int32_t*)c->d_fp, (const double*)c->d_Kvals, d_x, d_y, d_partials, done,     \
        (int32_t)c->spmv_keep_permille, c->nn, c->fp_cap, (int32_t)(c->spmv_rot == 64 ? 19 : c->spmv_rot), spmv_perm, c->spmv_perm_rounds
#define SPMV_FP_LAUNCH_NT(DM_, WPS_, NT_)                                                                        \
    do {                                                                                                        \
        if (lds > 48 * 1024)                                                                                    \
            FEMCY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmv_fp<DM_, WPS_, NT_>),            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
        if (ev)                                                                                                 \
            hipExtLaunchKernelGGL((k_spmv_fp<DM_, WPS_, NT_>), dim3(grid), dim3(BS), lds, c->stream, ea, eb, 0, SPMV_FP_ARGS); \
        else                                                                                                    \
            hipLaunchKernelGGL((k_spmv_fp<DM_, WPS_, NT_>), dim3(grid), dim3(BS), lds, c->stream, SPMV_FP_ARGS); \
    } while (0)
#define SPMV_FP_LAUNCH(DM_, WPS_)                              \
    do {                                                       \
        if (c->spmv_nt) SPMV_FP_LAUNCH_NT(DM_, WPS_, true);    \
        else SPMV_FP_LAUNCH_NT(DM_, WPS_, false);              \
    } while (0)
            if (c->dm == 3) {
                if (wps == 1) SPMV_FP_LAUNCH(3, 1); else if (wps == 2) SPMV_FP_LAUNCH(3, 2); else SPMV_FP_LAUNCH(3, 4);
            } else {
                if (wps == 1) SPMV_FP_LAUNCH(2, 1); else if (wps == 2) SPMV_FP_LAUNCH(2, 2); else SPMV_FP_LAUNCH(2, 4);
            }
#undef SPMV_FP_LAUNCH
#undef SPMV_FP_LAUNCH_NT
#undef SPMV_FP_ARGS
            FEMCY_HIP(hipGetLastError());
            if (nblocks_out) *nblocks_out = grid;
            return FEMCY_OK;
        }
    }
    if (c->dm == 3) {
        if (wps == 1) SPMV_LAUNCH(3, 1);
        else if (wps == 2) SPMV_LAUNCH(3, 2);
        else SPMV_LAUNCH(3, 4);
    } else {
        if (wps == 1) SPMV_LAUNCH(2, 1);
        else if (wps == 2) SPMV_LAUNCH(2, 2);
        else SPMV_LAUNCH(2, 4);
    }
#undef SPMV_LAUNCH
#undef SPMV_LAUNCH_NT
#undef SPMV_ARGS
    FEMCY_HIP(hipGetLastError());
    if (nblocks_out) *nblocks_out = grid;
    return FEMCY_OK;
}

int launch_spmv(Ctx* c, const double* d_x, double* d_y, double* d_partials, int* nblocks_out, bool pos_space) {
    return launch_spmv_impl(c, d_x, d_y, d_partials, nblocks_out, c->xcd, c->spmv_grid, nullptr, 0, pos_space);
}

// `reps` products back to back on the context's stream between one pair of HIP events, on the PCG's own vectors (d -> Ad)
// in node order or in storage order: the launch-to-launch time of the product as the three-launch loop issues it,
// kernel + the ~1.5 us boundary between dependent launches.  (Dispatch-attached events on single launches inside a
// solve read ~5 us high: the profiled packet drains the pipeline -- round 4: 71.1 us against 65.8 us in rocprofv3's
// kernel trace of the same run.)
// the two storage-order vectors of a single-rank solve (right-hand side in, solution out) and the block columns as
// positions; padding lanes start as zeros and are only ever written with zeros
int ensure_pos_vectors(Ctx* c) {
    int rc = ensure_bcolp(c);
    if (rc) return rc;
    const int64_t need = (int64_t)c->nslices * SLICE * c->dm + 64;
    if (c->pos_cap < need) {
        if (c->d_posb) (void)hipFree(c->d_posb);
        if (c->d_posx) (void)hipFree(c->d_posx);
        c->d_posb = c->d_posx = nullptr;
        c->pos_cap = 0;
        FEMCY_HIP(dmalloc(&c->d_posb, sizeof(double) * need));
        FEMCY_HIP(dmalloc(&c->d_posx, sizeof(double) * need));
        FEMCY_HIP(hipMemsetAsync(c->d_posb, 0, sizeof(double) * need, c->stream));
        FEMCY_HIP(hipMemsetAsync(c->d_posx, 0, sizeof(double) * need, c->stream));
        c->pos_cap = need;
        pcg_graph_reset(c);
    }
    return FEMCY_OK;
}

// the public product (femcy_spmv) through the storage-order kernel: permute x, multiply, permute back.  Pays when the
// internal row order is NOT the caller's numbering (FEMCY_OPT_NODE_ORDER chose a coordinate order because the caller's
// gathers badly: the C3D10 plates) and the vectors are large: 995 k C3D10 / 4.18 M DOF 868 -> ~620 us per call (round 6);
// equal at 548 k DOF (79 us), a loss where the caller's numbering IS the internal order (8 M C3D4: 281 -> 329 us, round 5)
int spmv_public_storage_order(Ctx* c, const double* d_x, double* d_y) {
    int rc = ensure_pos_vectors(c);
    if (rc) return rc;
    const int32_t npos = c->nslices * SLICE;
    const int pg = (npos + BS - 1) / BS;
    if (c->dm == 3) hipLaunchKernelGGL((k_to_pos<3>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, d_x, c->d_posb);
    else hipLaunchKernelGGL((k_to_pos<2>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, d_x, c->d_posb);
    if ((rc = launch_spmv(c, c->d_posb, c->d_posx, nullptr, nullptr, true))) return rc;
    if (c->dm == 3) hipLaunchKernelGGL((k_from_pos<3>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, (const double*)c->d_posx, d_y);
    else hipLaunchKernelGGL((k_from_pos<2>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, (const double*)c->d_posx, d_y);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int probe_spmv(Ctx* c, int32_t reps, int32_t storage_order, double* us_per_launch) {
    FEMCY_REQUIRE(c->have_pattern, "femcy_build_pattern must come first");
    FEMCY_REQUIRE(reps >= 1 && reps <= 100000 && us_per_launch, "probe_spmv: reps 1..1e5");
    const bool pos = storage_order != 0;
    if (pos) {
        int rc = ensure_bcolp(c);
        if (rc) return rc;
    }
    // x = the Jacobi vector of the last solve (any finite data does; zero after femcy_set_mesh), y = Ad; the stop flag of
    // the last solve must not turn the launches into no-ops
    FEMCY_HIP(hipMemsetAsync(&c->d_state->done, 0, sizeof(int32_t), c->stream));
    hipEvent_t e0, e1;
    FEMCY_HIP(hipEventCreate(&e0));
    FEMCY_HIP(hipEventCreate(&e1));
    int rc = FEMCY_OK;
    const int keep = c->opt_timing;
    c->opt_timing = 0;
    for (int pass = 0; pass < 2 && !rc; ++pass) {                // 5 warm-up launches, then the timed batch
        const int n = pass ? reps : 5;
        if (pass) (void)hipEventRecord(e0, c->stream);
        for (int k = 0; k < n && !rc; ++k) rc = launch_spmv(c, c->d_M, c->d_Ad, c->d_part1, nullptr, pos);
    }
    c->opt_timing = keep;
    (void)hipEventRecord(e1, c->stream);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) {
        set_error("probe_spmv: stream synchronisation failed");
        rc = FEMCY_EHIP;
    }
    float ms = 0.f;
    if (!rc) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    *us_per_launch = (double)ms * 1e3 / reps;
    return FEMCY_OK;
}

// the slices that hold at least one interface node, then all others (ascending inside each half, so the interior
// half keeps the spatial order the XCD ranges rely on)
int split_prepare(Ctx* c) {
    if (c->split_ready) return FEMCY_OK;
    std::vector<uint8_t> is_if((size_t)c->nslices, 0);
    for (int32_t dof : c->h_iface_dof) is_if[c->h_pos[dof / c->dm] >> 6] = 1;
    std::vector<int32_t> list;
    list.reserve(c->nslices);
    for (int32_t s2 = 0; s2 < c->nslices; ++s2)
        if (is_if[s2]) list.push_back(s2);
    c->n_if_slices = (int32_t)list.size();
    for (int32_t s2 = 0; s2 < c->nslices; ++s2)
        if (!is_if[s2]) list.push_back(s2);
    if (c->d_split_list) (void)hipFree(c->d_split_list);
    c->d_split_list = nullptr;
    FEMCY_HIP(dmalloc(&c->d_split_list, std::max<size_t>(list.size(), 1) * sizeof(int32_t)));
    FEMCY_HIP(hipMemcpy(c->d_split_list, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!c->comm_stream) FEMCY_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    if (!c->ev_iface) FEMCY_HIP(hipEventCreateWithFlags(&c->ev_iface, hipEventDisableTiming));
    if (!c->ev_xchg) FEMCY_HIP(hipEventCreateWithFlags(&c->ev_xchg, hipEventDisableTiming));
    c->split_ready = true;
    return FEMCY_OK;
}

int launch_spmv_part(Ctx* c, int part, const double* d_x, double* d_y, double* d_partials, int part_off,
                     int* nblocks_out) {
    const int32_t p0 = part == 1 ? 0 : c->n_if_slices, p1 = part == 1 ? c->n_if_slices : c->nslices;
    if (nblocks_out) *nblocks_out = 0;
    if (p1 <= p0) return FEMCY_OK;
    const int spb = 4 / c->spmv_wps;                       // slices per workgroup
    XcdRanges xr;
    for (int k = 0; k <= NXCD; ++k) xr.start[k] = p0 + (int32_t)((int64_t)(p1 - p0) * k / NXCD);
    int per = 1;
    for (int k = 0; k < NXCD; ++k) per = std::max(per, (xr.start[k + 1] - xr.start[k] + spb - 1) / spb);
    // the two halves together must fit the partial array: each half gets at most half of it
    const int grid = std::min(spmv_bpx(per, c->spmv_bpx_cap, c->spmv_cap_auto), MAX_PARTIALS / (2 * NXCD)) * NXCD;
    return launch_spmv_impl(c, d_x, d_y, d_partials ? d_partials + part_off : nullptr, nblocks_out, xr, grid,
                            c->d_split_list, d_partials ? part_off : 0);
}

// sum a sub-assembled vector over the ranks that share each interface DOF
static int iface_sum_p2p(Ctx* c, double* d_v, hipStream_t stream) {
    const int32_t total = c->h_nb_ptr.empty() ? 0 : c->h_nb_ptr.back();
    if (total > 0)
        hipLaunchKernelGGL(k_p2p_pack, dim3((total + BS - 1) / BS), dim3(BS), 0, stream, total, c->d_nb_dofs,
                           (const double*)d_v, c->d_nb_send);
    int rc = comm_neighbour_exchange(c, stream);
    if (rc) return rc;
    if (c->niface_local > 0)
        hipLaunchKernelGGL(k_p2p_sum, dim3((c->niface_local + BS - 1) / BS), dim3(BS), 0, stream, c->niface_local,
                           c->d_iface_dof, c->d_if_ptr, c->d_if_src, (const double*)c->d_nb_recv, d_v);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

int iface_sum(Ctx* c, double* d_v) {
    if (!c->comm) return FEMCY_OK;
    if (c->exchange == 1) return iface_sum_p2p(c, d_v, c->stream);
    hipLaunchKernelGGL(k_iface_pack_all, dim3((c->niface_global + BS - 1) / BS + 1), dim3(BS), 0, c->stream,
                       c->niface_global, c->d_slot2dof, d_v, c->d_commbuf, 0, (const double*)nullptr);
    int rc = comm_allreduce_sum(c, c->d_commbuf, c->niface_global);
    if (rc) return rc;
    if (c->niface_local > 0)
        hipLaunchKernelGGL(k_iface_unpack, dim3((c->niface_local + BS - 1) / BS), dim3(BS), 0, c->stream,
                           c->niface_local, c->d_iface_dof, c->d_iface_slot, c->d_commbuf, d_v);
    FEMCY_HIP(hipGetLastError());
    return FEMCY_OK;
}

// ------------------------------------------------------------------- small systems: one launch per solve
// Below ~1e4 DOF (every deck the reference ships) an iteration of the three-kernel loop costs 17 us -- launch latency,
// from a hipGraph as well -- for 1-2 us of work.  Here the whole solve is ONE persistent launch with ONE grid barrier
// per iteration: workgroup g multiplies slice g by d, publishes its rows of Ad (write-through stores) and its part of
// d.Ad, meets the others at the barrier, and then updates the WHOLE r and d -- redundantly, every workgroup the same
// arithmetic in the same order on its private copy (r and M in registers, d in LDS) -- so the reductions r.M.r and
// max|r| need no second exchange and the gathers of the next product read d from LDS.  x is kept in registers for the
// workgroup's own rows only and written once at the end.  Recurrence, preconditioner (Jacobi) and stopping rule are those of pcg_solve / the reference
// (conjugateGradientSolver.py:103-127).
// Inter-workgroup protocol (cdna_hip_programming.md, Guideline 16, form R1): payload = sc1 stores, every storing wave
// drains vmcnt, one lane arrives on a monotonic agent-scope counter; consumers poll that counter relaxed and read the
// payload with sc1 loads.  Ad and the d.Ad partials are double-buffered by iteration parity: a workgroup can run at
// most one barrier ahead of the slowest one.  Every spin is bounded; a timeout ends the launch with state.done = 3.
struct SmallPcg {
    const int32_t* slice_len;
    const int64_t* slice_off;
    const int32_t* bcol;
    const int32_t* node_of;
    const double* vals;
    const double* b;
    const double* M;
    double* x;
    double* Adbuf;        // [2][npad]
    double* part;         // [2][G]
    unsigned int* counter;
    PcgState* st;
    int32_t n, npad, maxit;
    uint32_t spin_limit;   // polls of the grid barrier before a workgroup gives up and poisons the counter
    int32_t dbg;           // probe build only (FEMCY_PERSIST_PROBE, tools/small_breakdown.py): 32 no Ad loads, 64 no wait,
                           // 128 no Ad stores, 256 no product, 512 no vector update
    double eps;
};

__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// K = ceil(n / 256): thread t owns the entries t, t + 256, ... of r and M and keeps them in REGISTERS for the whole
// solve (one workgroup per CU: a wave may use the full register file); only d, which the product gathers, lives in LDS
// RR: the first RR block rows of each wave's share of the slice stay in REGISTERS for the whole solve as well (a wave's
// share is a quarter of the slice's block rows: the register file holds 8-16 of them next to r and M); the rest is streamed from
// L2 as before.  Keeping those in LDS as well was measured (same time: the iteration is bound by its synchronisation
// chain, not by the product) and dropped
#ifdef FEMCY_PERSIST_PROBE
#define SDBG(a_, bit_) ((a_).dbg & (bit_))
#else
#define SDBG(a_, bit_) false
#endif
template <int DM, int K, int RR>
__global__ void __launch_bounds__(BS) k_pcg_small(SmallPcg a) {
    extern __shared__ __attribute__((aligned(16))) double lds_small[];
    double* d_l = lds_small;
    double* red = d_l + a.npad;                 // [4][DM][64] partial rows of the four waves
    double* sm1 = red + 4 * DM * 64;
    double* sm2 = sm1 + BS / 64;
    __shared__ int s_fail;
    const int g = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    constexpr int DD = DM * DM, NP = DD / 2;
    if (tid == 0) s_fail = 0;

    // ---- x0 = 0, r = b, d = M r; r.M.r and max|r0| (every workgroup computes the same numbers)
    double rr[K], mm[K];
    double accs = 0.0, accm = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = tid + k * BS;
        const bool in = i < n;
        const int ic = in ? i : n - 1;
        const double bi = in ? a.b[ic] : 0.0, mi = a.M[ic];
        rr[k] = bi;
        mm[k] = mi;
        if (in) d_l[i] = mi * bi;
        accs += bi * mi * bi;
        accm = fmax(accm, nan_to_inf_abs(bi));
    }
    double rMr = block_sum(accs, sm1);
    const double r0 = block_max(accm, sm2);
    double rmax = r0;
    const int32_t node = a.node_of[(int64_t)g * SLICE + lane];   // row permutation (SELL-C-sigma); -1 = padding lane
    double xo[DM];                                               // the workgroup's own rows of x (wave 0)
#pragma unroll
    for (int r = 0; r < DM; ++r) xo[r] = 0.0;
    int done = (r0 == 0.0) ? 1 : ((r0 != r0 || isinf(r0)) ? 2 : 0);
    int it = 0;
    const int32_t L = a.slice_len[g];
    const int64_t off = a.slice_off[g];
    const int32_t chunk = (L + 3) / 4;
    const int32_t j0 = wave * chunk, j1 = min(L, j0 + chunk);
    const int32_t* __restrict__ bc = a.bcol + off * SLICE + lane;
    const double2* __restrict__ vp = reinterpret_cast<const double2*>(a.vals + off * (int64_t)(DD * SLICE)) + lane;
    const double* __restrict__ vs = a.vals + off * (int64_t)(DD * SLICE) + NP * (2 * SLICE) + lane;
    double rv[RR > 0 ? RR : 1][DD];
    int32_t rcl[RR > 0 ? RR : 1];
#pragma unroll
    for (int jj = 0; jj < RR; ++jj) {
        const bool has = j0 + jj < j1;
        const int32_t j = has ? j0 + jj : 0;
#pragma unroll
        for (int kp = 0; kp < NP; ++kp) {
            const double2 t = vp[(int64_t)j * (DD * SLICE / 2) + kp * SLICE];
            rv[jj][2 * kp] = has ? t.x : 0.0;
            rv[jj][2 * kp + 1] = has ? t.y : 0.0;
        }
        if (DD & 1) rv[jj][DD - 1] = has ? vs[(int64_t)j * (DD * SLICE)] : 0.0;
        rcl[jj] = has ? bc[(int64_t)j * SLICE] : 0;               // a row the wave does not have: zero block on node 0
    }
    while (!done && it < a.maxit) {
        __syncthreads();                                 // d_l of the previous iteration is complete
        // ---- rows of slice g of K d, four waves share the row (block columns j0 .. j1 each), d gathered from LDS
        double acc[DM];
#pragma unroll
        for (int r = 0; r < DM; ++r) acc[r] = 0.0;
#pragma unroll
        for (int jj = 0; jj < RR; ++jj) {
            if (SDBG(a, 256)) continue;
            double xv[DM];
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) xv[cc] = d_l[rcl[jj] * DM + cc];
#pragma unroll
            for (int r = 0; r < DM; ++r)
#pragma unroll
                for (int cc = 0; cc < DM; ++cc) acc[r] += rv[jj][r * DM + cc] * xv[cc];
        }
#pragma unroll 2
        for (int32_t j = j0 + RR; j < (SDBG(a, 256) ? 0 : j1); ++j) {
            const int32_t col = bc[(int64_t)j * SLICE];
            double xv[DM], e[DD];
#pragma unroll
            for (int kp = 0; kp < NP; ++kp) {
                const double2 t = vp[(int64_t)j * (DD * SLICE / 2) + kp * SLICE];
                e[2 * kp] = t.x;
                e[2 * kp + 1] = t.y;
            }
            if (DD & 1) e[DD - 1] = vs[(int64_t)j * (DD * SLICE)];
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) xv[cc] = d_l[col * DM + cc];
#pragma unroll
            for (int r = 0; r < DM; ++r)
#pragma unroll
                for (int cc = 0; cc < DM; ++cc) acc[r] += e[r * DM + cc] * xv[cc];
        }
#pragma unroll
        for (int r = 0; r < DM; ++r) red[(wave * DM + r) * 64 + lane] = acc[r];
        __syncthreads();
        double dot = 0.0;
        double dold[DM];
        double* Adw = a.Adbuf + (size_t)(it & 1) * a.npad;
        if (wave == 0 && node >= 0) {
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                double v = acc[r];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += red[(w * DM + r) * 64 + lane];
                if (!SDBG(a, 128)) st_sc1(Adw + (int64_t)node * DM + r, v);
                dold[r] = d_l[node * DM + r];
                dot += dold[r] * v;
            }
        }
        if (wave == 0) {                                 // only wave 0 holds rows: its own shuffle tree is the block sum
            const double pd = wave_sum(dot);
            if (lane == 0) st_sc1(a.part + (size_t)(it & 1) * G + g, pd);
        }
        // ---- grid barrier: arrive after the stores have drained, wait for everybody
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // bounded wait (a legitimate one is a few microseconds): the workgroup that times out poisons the counter,
            // which releases every other workgroup -- those spinning now and those that arrive later -- with the same
            // verdict (the scheme of k_pcg_persist's grid_barrier); the host then redoes the solve with the
            // three-kernel loop and does not try this kernel again on the context
            constexpr unsigned int POISON = 0x80000000u;
            __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int target = (unsigned int)G * (unsigned int)(it + 1);
            unsigned int spins = 0;
            for (;;) {
                const unsigned int v = __hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v & POISON) {
                    s_fail = 1;
                    break;
                }
                if (v >= target || SDBG(a, 64)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) {            // a workgroup is missing (not resident / died)
                    __hip_atomic_fetch_or(a.counter, POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_fail = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (s_fail) {
            done = 3;
            break;
        }
        // ---- alpha; x (own rows), r, reductions, d -- the whole vectors in every workgroup; all loads first
        const double ps = tid < G ? ld_sc1(a.part + (size_t)(it & 1) * G + tid) : 0.0;    // G <= 128 < BS
        double av[K];
#pragma unroll
        for (int k = 0; k < K; ++k) av[k] = SDBG(a, 32) ? mm[k] : ld_sc1(Adw + min(tid + k * BS, n - 1));
        const double dAd = SDBG(a, 64) ? 1.0 : block_sum(ps, sm1);
        const double alpha = rMr / dAd;
        if (wave == 0 && node >= 0) {
#pragma unroll
            for (int r = 0; r < DM; ++r) xo[r] += alpha * dold[r];
        }
        accs = 0.0;
        accm = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (SDBG(a, 512) && k > 0) continue;
            const bool in = tid + k * BS < n;
            const double ri = in ? rr[k] - alpha * av[k] : 0.0;
            rr[k] = ri;
            accs += ri * mm[k] * ri;
            accm = fmax(accm, nan_to_inf_abs(ri));
        }
        // r.M.r and max|r| through one pair of barriers
        accs = wave_sum(accs);
        accm = wave_max(accm);
        if (lane == 0) {
            sm1[wave] = accs;
            sm2[wave] = accm;
        }
        __syncthreads();
        const double rMr_new = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
        rmax = fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3]));
        ++it;
        if (!SDBG(a, 0xfe0) && (rmax != rmax || isinf(rmax) || rMr_new != rMr_new)) {
            done = 2;
        } else if (rmax < a.eps * r0) {
            done = 1;
        } else {
            const double beta = rMr_new / rMr;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (SDBG(a, 512) && k > 0) continue;
                const int i = tid + k * BS;
                if (i < n) d_l[i] = mm[k] * rr[k] + beta * d_l[i];
            }
        }
        rMr = rMr_new;
    }
    if (wave == 0 && node >= 0) {
#pragma unroll
        for (int r = 0; r < DM; ++r) a.x[(int64_t)node * DM + r] = xo[r];
    }
    if (g == 0 && tid == 0) {
        a.st->iters = it;
        a.st->r0 = r0;
        a.st->rmax = rmax;
        a.st->done = done;
        a.st->rMr[0] = rMr;
    }
}

// A hand-rolled grid barrier needs every workgroup of the launch resident at once.  The occupancy query answers for an
// otherwise idle device (MI355X_MICROARCH.md "Residency": the API may be one workgroup per CU high at 81-96 SGPRs, so
// a grid is accepted only with a margin of one per CU unless the kernel needs the whole CU anyway); a busy device is
// what the bounded spin + poison in the kernels is for.  FEMCY_TUNE_SKIP_OCCUPANCY_CHECK = 1 skips the query (tests of
// the time-out path).
bool coresident(Ctx* c, const void* fn, int block, size_t lds, int grid) {
    if (c->opt_skip_occupancy) return true;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (per_cu > 1) per_cu -= 1;                         // margin for the API's optimism
    return (int64_t)per_cu * c->persist_cus >= grid;
}

// eligibility + launch; returns FEMCY_OK with *handled = false when the system does not qualify
static int pcg_small_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, bool* handled) {
    *handled = false;
    constexpr int KMAX = 48;                               // 48 * 256 = 12 288 DOF (K = 80 spills the register file)
    if (!c->opt_small || c->small_failed || c->comm || c->opt_timing || c->nslices > c->small_max_wg ||
        c->n > (int64_t)KMAX * BS)
        return FEMCY_OK;
    const int64_t npad = (c->n + 1) & ~(int64_t)1;
    const size_t lds = (size_t)(npad + 4 * c->dm * 64 + 2 * (BS / 64)) * sizeof(double);
    if (lds + 256 > (size_t)c->small_max_lds) return FEMCY_OK;    // + the kernel's static LDS
    const int G = c->nslices;
    if (!c->d_small || c->small_cap < 2 * npad + 2 * G + 8) {
        if (c->d_small) (void)hipFree(c->d_small);
        c->d_small = nullptr;
        c->small_cap = 2 * npad + 2 * G + 8;
        FEMCY_HIP(dmalloc(&c->d_small, sizeof(double) * c->small_cap));
    }
    SmallPcg a;
    a.slice_len = c->d_slice_len; a.slice_off = c->d_slice_off; a.bcol = c->d_bcol; a.node_of = c->d_node_of;
    a.vals = c->d_Kvals; a.b = d_b; a.M = c->d_M; a.x = d_x;
    a.Adbuf = c->d_small;
    a.part = c->d_small + 2 * npad;
    a.counter = reinterpret_cast<unsigned int*>(c->d_small + 2 * npad + 2 * G);
    a.st = c->d_state;
    a.n = (int32_t)c->n; a.npad = (int32_t)npad; a.maxit = maxit; a.eps = eps;
    a.spin_limit = c->barrier_spin_limit;
    a.dbg = c->opt_persist_dbg;
    FEMCY_HIP(hipMemsetAsync(a.counter, 0, 8, c->stream));
    const int kneed = (int)((c->n + BS - 1) / BS);
    // register buckets: thread t keeps entries t + 256 k, k < K, of r and M
#define FEMCY_SMALL(DM_, K_, RR_)                                                                                  \
    do {                                                                                                           \
        const void* fn = reinterpret_cast<const void*>(&k_pcg_small<DM_, K_, RR_>);                                \
        if (lds > 48 * 1024)                                                                                       \
            FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
        if (!coresident(c, fn, BS, lds, G)) return FEMCY_OK;   /* not all workgroups fit at once: three launches */ \
        hipLaunchKernelGGL((k_pcg_small<DM_, K_, RR_>), dim3(G), dim3(BS), lds, c->stream, a);                     \
    } while (0)
    // register-resident block rows per wave: a wave's share of the longest slice, in buckets the register file holds
    // next to the K entries of r and M (19 registers per 3 x 3 block row)
    const int share = c->opt_small_rr < 0 ? ((int)c->max_row_blocks + 3) / 4 : c->opt_small_rr;
#define FEMCY_SMALL_K(DM_)                                                                       \
    do {                                                                                         \
        if (share == 0) { if (kneed <= 8) FEMCY_SMALL(DM_, 8, 0); else if (kneed <= 24) FEMCY_SMALL(DM_, 24, 0); else FEMCY_SMALL(DM_, 48, 0); } \
        else if (kneed <= 8) { if (share <= 8) FEMCY_SMALL(DM_, 8, 8); else FEMCY_SMALL(DM_, 8, 16); }  \
        else if (kneed <= 24) { if (share <= 8) FEMCY_SMALL(DM_, 24, 8); else FEMCY_SMALL(DM_, 24, 12); } \
        else FEMCY_SMALL(DM_, 48, 0);   /* 192 registers of r and M: no room for block rows */   \
    } while (0)
    if (c->dm == 3) FEMCY_SMALL_K(3); else FEMCY_SMALL_K(2);
#undef FEMCY_SMALL_K
#undef FEMCY_SMALL
    FEMCY_HIP(hipGetLastError());
    *handled = true;
    return FEMCY_OK;
}

void pcg_graph_reset(Ctx* c) {
    if (c->pcg_graph) (void)hipGraphExecDestroy(c->pcg_graph);
    c->pcg_graph = nullptr;
    c->pcg_graph_x = nullptr;
}

// ----------------------------------------------------------------------------------------- PCG
int pcg_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, int32_t* iters, double* r0,
              double* rmax) {
    const bool multi = c->comm != nullptr;   // a 1-rank communicator still runs the exchange path (testable on one GPU)
    size_t th = timing_begin(c, T_PCG);

    // M = 1 / diag(K) (M_init).  Multi-rank: K is sub-assembled, so diag is summed over the interface first.
    const int jg = (c->nn + BS - 1) / BS;
    if (c->dm == 3)
        hipLaunchKernelGGL((k_jacobi<3>), dim3(jg), dim3(BS), 0, c->stream, c->nn, c->d_pos, c->d_slice_off, c->d_Kvals,
                           c->d_M, multi ? 0 : 1);
    else
        hipLaunchKernelGGL((k_jacobi<2>), dim3(jg), dim3(BS), 0, c->stream, c->nn, c->d_pos, c->d_slice_off, c->d_Kvals,
                           c->d_M, multi ? 0 : 1);
    if (multi) {
        int rc = iface_sum(c, c->d_M);
        if (rc) return rc;
        hipLaunchKernelGGL(k_recip, dim3(ew_grid(c, c->n)), dim3(BS), 0, c->stream, c->n, c->d_M);
    }
    // one-launch forms: the small-system kernel (single rank), the persistent kernel (single rank; across ranks once
    // every rank agreed -- femcy_comm_persist_agree -- with its own agreement on the outcome inside pcg_persist_solve)
    {
        bool handled = false, persist = false;
        int rc = FEMCY_OK;
        if (!multi && (rc = pcg_small_solve(c, d_b, d_x, eps, maxit, &handled))) return rc;
        if (!handled) {
            if ((rc = pcg_persist_solve(c, d_b, d_x, eps, maxit, &handled))) return rc;
            persist = handled;
        }
        if (handled && !multi) {
            FEMCY_HIP(hipMemcpyAsync(c->h_state, c->d_state, sizeof(PcgState), hipMemcpyDeviceToHost, c->stream));
            FEMCY_HIP(hipStreamSynchronize(c->stream));
            // a grid-barrier time-out (a workgroup was not resident: shared GPU, CU mask, another persistent kernel): every
            // workgroup left with the same verdict; the solve is redone by the three-kernel loop below, which
            // re-initialises x, r and d, and the context does not try the one-launch form again
            if (c->h_state->done == 3) {
                if (persist) c->persist_failed = true; else c->small_failed = true;
                c->timing.barrier_timeouts++;
                handled = persist = false;
            }
        }
        if (handled) {
            timing_end(c, th);
            if (iters) *iters = c->h_state->iters;
            if (r0) *r0 = c->h_state->r0;
            if (rmax) *rmax = c->h_state->rmax;
            c->timing.pcg_iters += c->h_state->iters;
            if (persist) {
                c->timing.solves_persist++;
                if (c->opt_timing) c->timing.persist_iters += c->h_state->iters;
            } else {
                c->timing.solves_small++;
            }
            if (c->h_state->done == 2) {
                set_error("PCG breakdown: NaN/Inf residual after %d iterations (r0 = %g)", c->h_state->iters, c->h_state->r0);
                return FEMCY_ENUMERIC;
            }
            return FEMCY_OK;
        }
    }
    // ---- three launches per iteration.  Single rank: the loop runs in STORAGE order (opt_pos_space) -- b is permuted
    // once, M is taken straight from the diagonal blocks in storage order, x is permuted back at the end; every kernel
    // in between is element-wise or the product, whose gathers then follow bcolp.  Multi-rank keeps node order (its
    // interface lists and owner mask are DOF-indexed).
    const bool pos = !multi && c->opt_pos_space != 0;
    const int32_t npos = c->nslices * SLICE;
    const double* vb = d_b;
    double* vx = d_x;
    if (pos) {
        int rc = ensure_pos_vectors(c);
        if (rc) return rc;
        const int pg = (npos + BS - 1) / BS;
        if (c->dm == 3) {
            hipLaunchKernelGGL((k_jacobi_pos<3>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, c->d_slice_off, c->d_Kvals, c->d_M);
            hipLaunchKernelGGL((k_to_pos<3>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, d_b, c->d_posb);
        } else {
            hipLaunchKernelGGL((k_jacobi_pos<2>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, c->d_slice_off, c->d_Kvals, c->d_M);
            hipLaunchKernelGGL((k_to_pos<2>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, d_b, c->d_posb);
        }
        vb = c->d_posb;
        vx = c->d_posx;
    }
    const int64_t n2 = pos ? (int64_t)npos * c->dm / 2 : (c->n + 1) / 2;
    // element ranges of the vector kernels = the ranges whose matrix rows each XCD multiplies (c->xcd, in slices of 64
    // rows), in double2 units; exact in storage order, up to one sigma-window per boundary in node order; grid = 8 x
    // workgroups per XCD
    XcdRanges er;
    for (int k = 0; k <= NXCD; ++k)
        er.start[k] = (int32_t)std::min<int64_t>(n2, ((int64_t)c->xcd.start[k] * SLICE * c->dm + 1) / 2);
    er.start[0] = 0;
    er.start[NXCD] = (int32_t)n2;
    int64_t emax = 1;
    for (int k = 0; k < NXCD; ++k) emax = std::max<int64_t>(emax, er.start[k + 1] - er.start[k]);
    const int g = NXCD * (int)std::max<int64_t>(1, std::min<int64_t>((emax + BS - 1) / BS, std::max(1, c->ew_cap / NXCD)));
    // one vector kernel per iteration (k_update_fused) when every thread can keep its share in registers (<= 8 double2)
    // and all workgroups are resident; otherwise -- and across ranks, where a collective sits between the two -- the
    // two kernels
    int fused_g = 0, fused_u = 0;
    if (!multi && c->opt_fused_update && !c->fused_failed) {
        const int64_t bpx_min = (emax + (int64_t)BS * 8 - 1) / ((int64_t)BS * 8);
        const int64_t bpx = std::max<int64_t>(bpx_min, g / NXCD);
        if (bpx <= 128) {
            const int64_t per = (emax + bpx * BS - 1) / (bpx * BS);
            fused_u = per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : 8));
            fused_g = (int)bpx * NXCD;
            const void* fn = nullptr;
#define FEMCY_FUSED_FN(NT_) (fused_u == 1 ? (const void*)&k_update_fused<NT_, 1> : fused_u == 2 ? (const void*)&k_update_fused<NT_, 2> \
                             : fused_u == 4 ? (const void*)&k_update_fused<NT_, 4> : (const void*)&k_update_fused<NT_, 8>)
            fn = c->vec_nt ? FEMCY_FUSED_FN(true) : FEMCY_FUSED_FN(false);
#undef FEMCY_FUSED_FN
            if (!coresident(c, fn, BS, 0, fused_g)) fused_g = 0;
        }
        if (fused_g && !c->d_fused) {
            FEMCY_HIP(dmalloc(&c->d_fused, 1024 * 2 * 16));
            FEMCY_HIP(hipMemsetAsync(c->d_fused, 0, 1024 * 2 * 16, c->stream));
        }
    }
    const bool fused = fused_g > 0;
    hipLaunchKernelGGL(k_pcg_init, dim3(g), dim3(BS), 0, c->stream, n2, (const double2*)vb, (const double2*)c->d_M,
                       (double2*)vx, (double2*)c->d_r, (double2*)c->d_d, (const uint8_t*)(multi ? c->d_owner : nullptr),
                       c->d_part2);
    if (multi) {
        hipLaunchKernelGGL(k_sum_partials2, dim3(1), dim3(BS), 0, c->stream, g, c->d_part2, c->d_commbuf);
        int rc = comm_allgather(c, c->d_commbuf, c->d_gather, 2);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_pcg_init_final, dim3(1), dim3(BS), 0, c->stream, g, c->d_part2, c->d_state,
                       (const double*)(multi ? c->d_gather : nullptr), (int)c->nranks, eps);
    FEMCY_HIP(hipGetLastError());

    if (multi && c->exchange == 1 && c->opt_overlap) {
        int rc = split_prepare(c);
        if (rc) return rc;
    }
    // one CG iteration = 3 launches (+ the exchange in multi-rank mode); nothing in their arguments depends on
    // the iteration number or on eps (both live in PcgState), so a burst can be captured once and replayed
    auto enqueue_iteration = [&]() -> int {
        int np1 = 0;
        int rc;
        const double* dAd_red = nullptr;
        if (multi && c->exchange == 1 && c->opt_overlap) {
            // overlapped schedule: interface slices first; their exchange (pack, send/recv, rank-ordered sum into the
            // interface rows of Ad) runs on the comm stream while the main stream multiplies the interior slices; the
            // scalar d.K_loc.d needs both halves and travels in an 8-byte all-reduce of its own
            int gA = 0, gB = 0;
            if ((rc = launch_spmv_part(c, 1, c->d_d, c->d_Ad, c->d_part1, 0, &gA))) return rc;
            FEMCY_HIP(hipEventRecord(c->ev_iface, c->stream));
            if ((rc = launch_spmv_part(c, 2, c->d_d, c->d_Ad, c->d_part1, gA, &gB))) return rc;
            np1 = gA + gB;
            FEMCY_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_iface, 0));
            if ((rc = iface_sum_p2p(c, c->d_Ad, c->comm_stream))) return rc;
            FEMCY_HIP(hipEventRecord(c->ev_xchg, c->comm_stream));
            double* slot = c->d_commbuf + c->niface_global;
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(BS), 0, c->stream, np1, (const double*)c->d_part1, 0, slot);
            if ((rc = comm_allreduce_sum(c, slot, 1))) return rc;
            FEMCY_HIP(hipStreamWaitEvent(c->stream, c->ev_xchg, 0));
            dAd_red = slot;
        } else if ((rc = launch_spmv(c, c->d_d, c->d_Ad, c->d_part1, &np1, pos))) {
            return rc;
        } else if (multi && c->exchange == 1) {
            // neighbour send/recv of the interface entries of Ad, then the scalar d.Ad by an 8-byte all-reduce
            double* slot = c->d_commbuf + c->niface_global;
            if ((rc = iface_sum_p2p(c, c->d_Ad, c->stream))) return rc;
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(BS), 0, c->stream, np1, (const double*)c->d_part1, 0, slot);
            if ((rc = comm_allreduce_sum(c, slot, 1))) return rc;
            dAd_red = slot;
        } else if (multi) {
            // interface rows of Ad hold partial sums: pack them + the local d.Ad, all-reduce, unpack
            double* slot = c->d_commbuf + c->niface_global;
            hipLaunchKernelGGL(k_iface_pack_all, dim3((c->niface_global + BS - 1) / BS + 1), dim3(BS), 0, c->stream,
                               c->niface_global, c->d_slot2dof, c->d_Ad, c->d_commbuf, np1, c->d_part1);
            if ((rc = comm_allreduce_sum(c, c->d_commbuf, (int64_t)c->niface_global + 1))) return rc;
            if (c->niface_local > 0)
                hipLaunchKernelGGL(k_iface_unpack, dim3((c->niface_local + BS - 1) / BS), dim3(BS), 0, c->stream,
                                   c->niface_local, c->d_iface_dof, c->d_iface_slot, c->d_commbuf, c->d_Ad);
            dAd_red = slot;
        }
        if (fused) {
#define FEMCY_FU(NT_, U_)                                                                                          \
    hipLaunchKernelGGL((k_update_fused<NT_, U_>), dim3(fused_g), dim3(BS), 0, c->stream, er, np1, c->d_part1, c->d_state,  \
                       (const double2*)c->d_Ad, (const double2*)c->d_M, (double2*)c->d_r, (double2*)c->d_d, (double2*)vx, \
                       c->d_fused, c->barrier_spin_limit)
#define FEMCY_FU_U(NT_)                                                                                            \
    do {                                                                                                           \
        if (fused_u == 1) FEMCY_FU(NT_, 1); else if (fused_u == 2) FEMCY_FU(NT_, 2);                               \
        else if (fused_u == 4) FEMCY_FU(NT_, 4); else FEMCY_FU(NT_, 8);                                            \
    } while (0)
            if (c->vec_nt) FEMCY_FU_U(true); else FEMCY_FU_U(false);
#undef FEMCY_FU_U
#undef FEMCY_FU
            return FEMCY_OK;
        }
#define FEMCY_XR(NT_, MU_)                                                                                         \
    hipLaunchKernelGGL((k_update_xr<NT_, MU_>), dim3(g), dim3(BS), 0, c->stream, er, np1, c->d_part1, dAd_red,       \
                       c->d_state, (const double2*)c->d_Ad, (const double2*)c->d_M, (double2*)c->d_r,                \
                       (const uint8_t*)(multi ? c->d_owner : nullptr), c->d_part2)
        if (multi) { if (c->vec_nt) FEMCY_XR(true, true); else FEMCY_XR(false, true); }
        else       { if (c->vec_nt) FEMCY_XR(true, false); else FEMCY_XR(false, false); }
#undef FEMCY_XR
        if (multi) {
            double* pair = c->d_commbuf + c->niface_global + 2;
            hipLaunchKernelGGL(k_sum_partials2, dim3(1), dim3(BS), 0, c->stream, g, c->d_part2, pair);
            if ((rc = comm_allgather(c, pair, c->d_gather, 2))) return rc;
        }
#define FEMCY_UD(NT_)                                                                              \
    hipLaunchKernelGGL(k_update_d<NT_>, dim3(g), dim3(BS), 0, c->stream, er, g, c->d_part2,         \
                       (const double*)(multi ? c->d_gather : nullptr), (int)c->nranks, c->d_state, \
                       (const double2*)c->d_r, (const double2*)c->d_M, (double2*)c->d_d, (double2*)vx)
        if (c->vec_nt) FEMCY_UD(true); else FEMCY_UD(false);
#undef FEMCY_UD
        return FEMCY_OK;
    };

    // hipGraph of one poll-burst: removes the per-launch host cost, which dominates below ~1e5 DOF
    const int P = c->opt_poll;
    // measured on MI355X: replay is ~17 % faster end-to-end on a 969-DOF deck (launch-bound) and ~2 us per
    // iteration SLOWER at 548 535 DOF (GPU-bound), so "auto" (1) only uses it below 2e5 DOF; 2 forces it on
    const bool want_graph = c->opt_graph == 2 || (c->opt_graph == 1 && c->n < 200000);
    const bool use_graph = want_graph && !multi && !c->opt_timing && maxit >= P;
    if (use_graph && (!c->pcg_graph || c->pcg_graph_x != vx || c->pcg_graph_iters != P || c->pcg_graph_g != g + 4096 * fused_g ||
                      c->pcg_graph_np1 != c->spmv_grid)) {
        pcg_graph_reset(c);
        hipGraph_t graph = nullptr;
        FEMCY_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        int rc = FEMCY_OK;
        for (int k = 0; k < P && !rc; ++k) rc = enqueue_iteration();
        hipError_t ce = hipStreamEndCapture(c->stream, &graph);
        if (rc) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        FEMCY_HIP(ce);
        FEMCY_HIP(hipGraphInstantiate(&c->pcg_graph, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        c->pcg_graph_x = vx;
        c->pcg_graph_iters = P;
        c->pcg_graph_g = g + 4096 * fused_g;
        c->pcg_graph_np1 = c->spmv_grid;
    }

    int32_t it = 0;
    bool finished = false;
    while (!finished) {
        if (use_graph && (int64_t)it + P <= maxit) {
            FEMCY_HIP(hipGraphLaunch(c->pcg_graph, c->stream));
            it += P;
        } else {
            const int32_t burst_end = (int32_t)std::min<int64_t>((int64_t)it + P, maxit);
            for (; it < burst_end; ++it) {
                int rc = enqueue_iteration();
                if (rc) return rc;
            }
        }
        FEMCY_HIP(hipGetLastError());
        FEMCY_HIP(hipMemcpyAsync(c->h_state, c->d_state, sizeof(PcgState), hipMemcpyDeviceToHost, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        if (c->h_state->done || it >= maxit) finished = true;
    }
    if (fused && c->h_state->done == 3) {
        // the in-kernel exchange of k_update_fused timed out (a workgroup was not resident: shared GPU, CU mask): the
        // solve is redone with the two vector kernels, which this context keeps from now on
        c->fused_failed = true;
        c->timing.barrier_timeouts++;
        pcg_graph_reset(c);
        timing_end(c, th);
        return pcg_solve(c, d_b, d_x, eps, maxit, iters, r0, rmax);
    }
    if (pos) {
        const int pg = (npos + BS - 1) / BS;
        if (c->dm == 3) hipLaunchKernelGGL((k_from_pos<3>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, (const double*)c->d_posx, d_x);
        else hipLaunchKernelGGL((k_from_pos<2>), dim3(pg), dim3(BS), 0, c->stream, npos, c->d_node_of, (const double*)c->d_posx, d_x);
        FEMCY_HIP(hipGetLastError());
    }
    timing_end(c, th);
    if (iters) *iters = c->h_state->iters;
    if (r0) *r0 = c->h_state->r0;
    if (rmax) *rmax = c->h_state->rmax;
    c->timing.pcg_iters += c->h_state->iters;
    c->timing.solves_three++;
    if (c->h_state->done == 2) {
        if (getenv("FEMCY_DEBUG_DUMP_BREAKDOWN")) {              // where did the first non-finite entry appear?
            const int64_t nd = pos ? (int64_t)npos * c->dm : c->n;
            std::vector<double> h((size_t)nd);
            const struct { const char* name; const double* p; } vecs[] = {{"b", vb}, {"M", c->d_M}, {"r", c->d_r}, {"d", c->d_d}, {"Ad", c->d_Ad}, {"x", vx}};
            for (const auto& v : vecs) {
                (void)hipMemcpy(h.data(), v.p, sizeof(double) * nd, hipMemcpyDeviceToHost);
                int64_t bad = 0, first = -1, zeros = 0;
                for (int64_t i = 0; i < nd; ++i) {
                    if (!std::isfinite(h[i])) { if (first < 0) first = i; ++bad; }
                    if (h[i] == 0.0) ++zeros;
                }
                fprintf(stderr, "[femcy debug] %s: %lld of %lld non-finite (first at %lld = slice %lld lane %lld comp %lld), %lld zeros\n", v.name,
                        (long long)bad, (long long)nd, (long long)first, (long long)(first / (64 * c->dm)), (long long)(first / c->dm % 64),
                        (long long)(first % c->dm), (long long)zeros);
            }
            fprintf(stderr, "[femcy debug] state: iters %d r0 %g rmax %g dAd %g alpha %g rMr %g %g; pos %d npos %d n %lld g %d\n", c->h_state->iters,
                    c->h_state->r0, c->h_state->rmax, c->h_state->dAd, c->h_state->alpha, c->h_state->rMr[0], c->h_state->rMr[1], (int)pos, (int)npos,
                    (long long)c->n, g);
        }
        set_error("PCG breakdown: NaN/Inf residual after %d iterations (r0 = %g)", c->h_state->iters, c->h_state->r0);
        return FEMCY_ENUMERIC;
    }
    return FEMCY_OK;
}

}  // namespace femcy
