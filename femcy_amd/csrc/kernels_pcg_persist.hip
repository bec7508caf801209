// Jacobi-PCG as ONE persistent launch for single-rank systems that fit one wavefront-task per SIMD (<= 4 slices of
// 64 rows per wave: ~7e5 DOF on MI355X): the headline 1 M-element C3D4 configuration (matrix from the Infinity Cache)
// and, since round 5, the 124 k C3D10 plate whose 380 MB matrix streams from HBM (61 against 78 us per iteration with
// three launches).  DESIGN.md section 3 has the measurements.
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
//     next product) are grid-wide exchanges without fences (cdna_hip_programming.md Guideline 16, R1 / R2), in one of
//     two forms chosen per launch (FEMCY_TUNE_PERSIST_VARIANT, measured in DESIGN.md section 3):
//       - counters: per-XCD arrival counters + one top counter, relaxed agent-scope atomics, data exchanged with sc1
//         (write-through) stores and sc1 loads (round 2);
//       - tagged granules (VAR & 2): every workgroup publishes {value, round} as ONE 16-byte sc1 store and one wave per
//         workgroup sweeps all G granules until every tag shows the round -- barrier and exchange in one store
//         latency + one load latency instead of store-ack + two atomics + poll + data load.
// Round 3 variants (template parameter VAR, bits; measurements in DESIGN.md section 3):
//   1  the streamed block rows are loaded non-temporally, except the first `l2_rows` streamed rows of every slice, which
//      keep the default policy: the stream is latency-bound at one wave per SIMD (femcy_probe_stream: 7.4 TB/s from the
//      Infinity Cache in this launch shape against 24 TB/s at eight workgroups per CU), nt loads return sooner, and
//      the default-policy rows are what stays in the XCD's 4 MiB L2 from one product to the next;
//   2  tagged-granule synchronisation (above);
//   8  (round 4) in-band validity of the published d: sentinel words + triple buffer instead of the third exchange
//      (V_INBAND below); needs bit 4
//   4  d is published in STORAGE order (position = slice * 64 + lane) so that a lane's dm values are contiguous for
//      the whole wave: one 16-byte + one 8-byte store / gather per node instead of three 8-byte ones (a third fewer
//      texture-address cycles on the largest non-matrix item of the iteration).
// Measured and dropped in round 3: sweeping the streamed rows in alternating direction on odd / even iterations (the
// tail of one sweep re-read from L2 as the head of the next): +4.7 us per iteration -- two copies of the product loop
// spill, and the reuse does not materialise.
// Recurrence, preconditioner and stopping rule are those of pcg_solve / the reference
// (conjugateGradientSolver.py:103-127); partial sums are combined in a fixed order, so a solve is bit-reproducible.
// d is double-buffered by iteration parity (a wave may gather d_k while a faster one already publishes d_k+1), as are
// the partial arrays.  Every spin is bounded: the workgroup that times out poisons the exchange (which releases all
// others), the launch ends with state.done = 3 and the host falls back to the three-kernel loop.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>
#include "ctx.hpp"
#include "wave_reduce.hpp"
#include "granule.hpp"

// The inter-workgroup protocol below (sc1 write-through stores, sc1 loads, relaxed agent-scope counters, no fences)
// rests on the cache-policy semantics of the multi-XCD CDNA3 / CDNA4 parts (MI355X_MICROARCH.md "Workgroup dispatch,
// XCD placement & inter-workgroup visibility"); another --offload-arch must not silently lose coherence.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_pcg_persist: the sc1-only hand-off protocol is validated for gfx942 / gfx950 only"
#endif

namespace femcy {

namespace {

constexpr int PBS = 256;        // 4 waves per workgroup, one workgroup per CU
constexpr int PNX = 8;
constexpr int CH = 4;         // block rows per batch of the LDS-resident and the streamed part
constexpr int V_NT = 1, V_A2A = 2, V_WIDE = 4, V_INBAND = 8;
// V_INBAND: the published d validates itself.  A word of d that has not been written yet holds a sentinel (a NaN with a
// payload no arithmetic produces), the consumers' gathers ARE the poll, and the third grid-wide exchange of the
// iteration ("d published") is gone.  d is triple-buffered by iteration: at the end of iteration k a lane publishes
// d_(k+1) into buffer (k+1) % 3 and re-arms buffer (k+2) % 3 = (k-1) % 3 with the sentinel -- that buffer was last
// read by product k-1, which every wave had finished before anybody passed the first exchange of iteration k-1, and the
// re-arm has been acknowledged (in-order VMEM returns: it sits in front of product k+1's first load) before its owner
// arrives at the first exchange of iteration k+1, behind which the next publish into it lies.  A gather therefore
// sees either the sentinel or the value it wants, never an older one.  Each 8-byte word is written by one store
// instruction and lies inside one cache line (8-byte aligned), so it is never seen half-written.
constexpr unsigned SENT_HI = 0xFFFBADD0u, SENT_LO = 0x5E471E70u;

// Work-skipping switches for timing experiments (tools/persist_breakdown.py) exist only in a probe build
// (FEMCY_EXTRA_FLAGS=-DFEMCY_PERSIST_PROBE FEMCY_OUT=../libfemcy_hip_probe.so csrc/build.sh): the shipped library
// has no path that skips work.
#ifdef FEMCY_PERSIST_PROBE
#define PDBG(a_, bit_) ((a_).dbg & (bit_))
#else
#define PDBG(a_, bit_) false
#endif

struct PersistPcg {
    const int32_t* slice_len;
    const int64_t* slice_off;
    const int32_t* bcol;    // block columns as node numbers, or as storage positions (VAR & 4)
    const int32_t* node_of;
    const int32_t* assign;  // [8 * waves per XCD][SPW] slices of each wave, -1 = none
    const double* vals;
    const double* b;
    const double* M;
    double* x;
    double* dbuf;         // [2][npad]
    double* part1;        // [2][G]      d.Ad partials
    double* part2;        // [2][2 G]    (r.M.r, max|r|) pairs
    double* slots;        // tagged granules {value, tag}: [G] d.Ad, [2 G] (r.M.r, max|r|), [G] d published
    unsigned int* xc;     // [8][32]     per-XCD arrival counters (one cache line apart)
    unsigned int* top;
    PcgState* st;
    int32_t npad, maxit, lds_rows, dbg, l2_rows;
    uint32_t spin_limit;  // polls before a barrier gives up and poisons the exchange
    uint32_t xspin_limit; // the same for polls of another rank's words (ranks start their launches milliseconds apart)
    double eps;
    // ---- across ranks (template parameter MULTI): mailboxes written by the peers' kernels, see the block comment
    // "persistent PCG across ranks" below
    unsigned long long* mbox;                 // this rank's mailbox
    unsigned long long* const* peer;          // [nranks] every rank's mailbox as mapped on this device
    const int32_t* mr_tab;                    // [positions][4]: send entry, recv entry, neighbour | lower << 8, its nb_total
    const uint8_t* owner;                     // [n] 1 = this rank counts the DOF in reductions
    int32_t rank, nranks, nb_total;
    uint32_t tagbase;                         // solve serial << 20: tags never repeat from one solve to the next
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
        while (!PDBG(a, 8)) {
            const unsigned v = __hip_atomic_load(a.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v & POISON) {
                *s_fail = 1;
                break;
            }
            if (v >= PNX * (round + 1)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > a.spin_limit) {
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

// ---- tagged-granule exchange (wave 0 of every workgroup): granule.hpp

// ------------------------------------------------------------------------------- persistent PCG across ranks
// With a communicator attached round 2 fell back to three launches + 2-3 RCCL calls per iteration (51-63 us against
// 29 us on one GPU, before any real link latency).  Here every rank keeps its ONE launch per solve; what crosses
// the ranks travels through mailboxes -- one buffer per rank in its own HBM (fine-grained), mapped into the peers
// (hipIpc / peer access) and WRITTEN BY THE PEERS' KERNELS over xGMI, polled locally:
//   * entry = two 8-byte words {value lo32 | tag << 32, value hi32 | tag << 32}: every word validates itself (the
//     LL protocol of RCCL), so no flag, no fence and no ordering between the words is needed; 8-byte system-scope
//     relaxed atomics both sides;
//   * interface rows of Ad: the lane that owns a shared row writes its partial sum straight into the sharing rank's
//     mailbox right after the product and reads the neighbour's partial before the r update; the two are added in
//     ascending rank order, so the replicas stay bit-identical (z-slabs: one sharer per interface node -- the form
//     this path takes; other partitions keep the RCCL loop);
//   * d.Ad and (r.M.r, max|r|): local grid-wide exchange first, then workgroup 0 writes the rank's value into every
//     rank's mailbox and wave 0 of every workgroup polls the nranks entries of its own rank's mailbox and combines
//     them in rank order.  d.Ad = sum_r d_r.K_r d_r needs no owner mask; r.M.r counts a shared DOF on its owner;
//   * entries are double-buffered by iteration parity and tagged (solve serial << 20) | (iteration mod 2^20), so nothing
//     is ever re-armed; a rank cannot run more than one exchange ahead of a peer (it needs the peer's value to pass).
// Every poll is bounded (spin_limit); a time-out ends the launch with done = 3 and the host -- after agreeing with
// the other ranks through the communicator -- redoes the solve with the RCCL loop.
constexpr int MB_SA = 0;                                         // [2][R] entries: d.Ad
__device__ __forceinline__ int mb_sb(int R) { return 4 * R; }    // [2][R][2] entries: (r.M.r, max|r|)
__device__ __forceinline__ int mb_ad(int R) { return 12 * R; }   // [2][nb_total] entries: interface rows of Ad
// tag of the exchanges of iteration `it` (-1 = the set-up exchange): solve serial in the upper 12 bits, the iteration
// count wraps inside the lower 20 -- an entry is rewritten every second iteration (parity double buffer), so a wrapped
// tag can only meet the entry of two iterations earlier, never an equal one; never 0 (the zeroed mailbox)
__device__ __forceinline__ uint32_t mb_tag(uint32_t tagbase, int it) {
    return tagbase | ((uint32_t)(it + 1) & 0xFFFFFu);
}
__device__ __forceinline__ void mb_store(unsigned long long* p, double v, uint32_t tag) {
    const unsigned long long t = (unsigned long long)tag << 32;
    __hip_atomic_store(p, t | (uint32_t)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p + 1, t | (uint32_t)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool mb_load(const unsigned long long* p, uint32_t tag, double& v) {
    const unsigned long long w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v = __hiloint2double((int)(uint32_t)w1, (int)(uint32_t)w0);
    return (uint32_t)(w0 >> 32) == tag && (uint32_t)(w1 >> 32) == tag;
}
// wave 0 of a workgroup: NV values of this rank (the same in every lane) -> combined over the ranks in rank order.
// `base` = word offset of the exchange's area, entries [parity][rank][NV].  Workgroup 0 is the sender.
template <int NV>
__device__ __forceinline__ bool xrank_reduce(const PersistPcg& a, int base, int parity, uint32_t tag, double (&val)[NV],
                                             const int (&op)[NV]) {
    const int lane = threadIdx.x & 63, R = a.nranks;
    // (round 5) a rank's own value does not travel through its mailbox: lane `rank` takes it from the register.  The
    // store -> visible -> load round trip on the own fine-grained buffer cost ~1.3 us per exchange, 2.6 us per iteration
    // (32.2 against 27.0 us with a 1-rank communicator); the values and their order of summation are unchanged
    if (blockIdx.x == 0 && lane < R && lane != a.rank) {
        unsigned long long* dst = a.peer[lane] + base + (size_t)((parity * R + a.rank) * NV) * 2;
#pragma unroll
        for (int v = 0; v < NV; ++v) mb_store(dst + 2 * v, val[v], tag);
    }
    double got[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) got[v] = lane == a.rank ? val[v] : 0.0;
    uint32_t spins = 0;
    for (;;) {
        bool ok = true;
        if (lane < R && lane != a.rank) {
            const unsigned long long* src = a.mbox + base + (size_t)((parity * R + lane) * NV) * 2;
#pragma unroll
            for (int v = 0; v < NV; ++v) ok = mb_load(src + 2 * v, tag, got[v]) && ok;
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > a.xspin_limit) return false;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double acc = 0.0;
        for (int r = 0; r < R; ++r) {
            const double x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(got[v]), r),
                                              __builtin_amdgcn_readlane(__double2loint(got[v]), r));
            acc = op[v] ? fmax(acc, x) : acc + x;
        }
        val[v] = acc;
    }
    return true;
}

// V_INBAND: after a batch of gathers, repeat it until no lane holds a sentinel word (the producer has not published
// yet); wave-uniform loop, bounded; the "memory" clobber keeps the compiler from re-using the first loads
#define FEMCY_SETTLE(NB_, REGATHER_)                                                     \
    if (INB) {                                                                           \
        uint32_t spins_ = 0;                                                             \
        for (;;) {                                                                       \
            bool st_ = false;                                                            \
            _Pragma("unroll") for (int u_ = 0; u_ < (NB_); ++u_) st_ = st_ || stale_d(xg[u_]); \
            if (!__any(st_)) break;                                                      \
            if (++spins_ > a.spin_limit) {                                               \
                pfail = true;                                                            \
                break;                                                                   \
            }                                                                            \
            __builtin_amdgcn_s_sleep(1);                                                 \
            asm volatile("" ::: "memory");                                               \
            REGATHER_;                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                           \
        }                                                                                \
    }

template <int DM, int SPW, int RJ, int VAR, bool MULTI = false>
__global__ void __launch_bounds__(PBS) k_pcg_persist(PersistPcg a) {
    constexpr int DD = DM * DM, NP = DD / 2;
    constexpr bool NT = (VAR & V_NT) != 0, A2A = (VAR & V_A2A) != 0, WIDE = (VAR & V_WIDE) != 0;
    constexpr bool INB = (VAR & V_INBAND) != 0;
#ifndef FEMCY_PERSIST_OWN_DIAG
#define FEMCY_PERSIST_OWN_DIAG 1
#endif
    // block row 0 of a slice is the diagonal block: its column is the lane's own d, which is in registers -- no gather
    // for it (round 4: 27.30 -> 27.07 us per iteration at 1 M C3D4, bit-identical iterates; profiles/r04_persist_inband.txt)
    constexpr bool OWN_DIAG = INB || (FEMCY_PERSIST_OWN_DIAG && WIDE);
    static_assert(!INB || (WIDE && A2A), "in-band validity of d is built on the storage-order, tagged-granule form");
    constexpr int NDB = INB ? 3 : 2;                             // buffers of the published d
    extern __shared__ __attribute__((aligned(16))) char lds_persist[];
    __shared__ double sm1[PBS / 64], sm2[PBS / 64], bc[2];
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
    int32_t sl[SPW], Ls[SPW], node[SPW], dpos[SPW];
    uint32_t inmask = 0;                                          // bit t: this lane is a node of slice t (not a padding lane)
#define IN(t_) (((inmask >> (t_)) & 1u) != 0)
    // per-slice scalars are kept SMALL: first stored block row as 32 bits (femcy_build_pattern refuses more than 2^31 stored
    // blocks), the three row marks packed into one word.  With seven scalar words per slice the 6- and 7-slice shapes of
    // 3 x 3 blocks needed 77 / 110 scalar spill slots -- more than the 64 lanes of one spill register -- and faulted on
    // the first launch ("write access to a read-only page", every mesh size; 54 slots at five slices run): round 6
    int32_t offs32[SPW];
#define OFFS(t_) ((int64_t)offs32[t_])
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
        const int32_t s = __builtin_amdgcn_readfirstlane(a.assign[((size_t)xk * nwx + wx) * SPW + t]);
        const bool act = s >= 0;
        sl[t] = act ? s : -1;
        Ls[t] = act ? __builtin_amdgcn_readfirstlane(a.slice_len[s]) : 0;
        const int64_t o = act ? a.slice_off[s] : 0;
        offs32[t] = __builtin_amdgcn_readfirstlane((int)(uint32_t)o);
        node[t] = act ? a.node_of[(int64_t)s * 64 + lane] : -1;
        // (round 6: `node` lives until the vectors are initialised and is re-loaded for the final store of x -- the loop
        // tests the bit.  With 6 / 7 slices of 3 x 3 blocks per wave hipcc 7.0 kept node[t] in an accumulation register it
        // also handed to a double: the final store went through garbage, profiles/r06_persist_spw67_fault.txt)
        if (node[t] >= 0) inmask |= 1u << t;
        // where this lane's d lives in the published vector: its node (round 2), or its storage position (VAR & 4)
        dpos[t] = WIDE ? (act ? s * 64 + lane : 0) : (node[t] >= 0 ? node[t] : 0);
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
            const double* src = a.vals + (OFFS(t) + (has ? jj : 0)) * (int64_t)(DD * 64);
#pragma unroll
            for (int k = 0; k < DD; ++k) rv[t][jj][k] = has ? src[kv_index<DM>(0, k, lane)] : 0.0;
            rcl[t][jj] = has ? a.bcol[(OFFS(t) + jj) * 64 + lane] : dpos[t];
        }
    // LDS rows are handed out from the LAST slice backwards, so that slice 0 keeps streamed rows: its first batch is
    // the one prefetched during the synchronisation windows (below).  JL(t) .. JE(t)-1 = the slice's rows in LDS.
    uint32_t jq[SPW];                                             // jl | ql << 8 | je << 16
#define JL(t_) ((int32_t)(jq[t_] & 0xffu))
#define QL(t_) ((int32_t)((jq[t_] >> 8) & 0xffu))
#define JE(t_) ((int32_t)(jq[t_] >> 16))
    {
        int qn = 0;
#pragma unroll
        for (int t = SPW - 1; t >= 0; --t) {
            const int jl_ = min(RJ, Ls[t]);
            const int nl = max(0, min(Ls[t] - jl_, a.lds_rows - qn));
            jq[t] = (uint32_t)jl_ | ((uint32_t)qn << 8) | ((uint32_t)(jl_ + nl) << 16);
            qn += nl;
            for (int32_t j = JL(t); j < JE(t); ++j) {
                const int q = QL(t) + (j - JL(t));
                const double* src = a.vals + (OFFS(t) + j) * (int64_t)(DD * 64);
#pragma unroll
                for (int k = 0; k < DD; ++k) lvals[(q * DD + k) * 64 + lane] = src[kv_index<DM>(0, k, lane)];
                lcols[q * 64 + lane] = a.bcol[(OFFS(t) + j) * 64 + lane];
            }
        }
    }
    // d is gathered with sc1 BUFFER loads: the same cache policy as an agent-scope atomic load (the other XCDs wrote d
    // with sc1 stores), but an ordinary load to the compiler, which may then issue the gathers of several block rows
    // before the first wait (with atomic loads it serialised them: 15 + 8 dependent L2 round trips per product)
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dbuf, 0, (int)((size_t)NDB * a.npad * sizeof(double)), 0x00020000);
    auto gather_d = [&](int32_t col, int32_t parity_off, double (&xv)[DM]) {
        if (WIDE) {     // dm contiguous doubles at 8-byte alignment: 16 B (+ 8 B for dm = 3); dword alignment suffices
            const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(drsrc, col * (DM * 8), parity_off, AUX_SC1);
            xv[0] = __hiloint2double((int)w.y, (int)w.x);
            xv[1] = __hiloint2double((int)w.w, (int)w.z);
            if (DM == 3) {
                const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(drsrc, col * (DM * 8) + 16, parity_off, AUX_SC1);
                xv[DM - 1] = __hiloint2double((int)w2.y, (int)w2.x);
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(drsrc, (col * DM + cc) * 8, parity_off, AUX_SC1);
                xv[cc] = __hiloint2double((int)w.y, (int)w.x);
            }
        }
    };
    auto publish_d = [&](int32_t p, int32_t parity_off, const double (&dv)[DM]) {
        if (WIDE) {
            u32x4 w;
            w.x = (unsigned)__double2loint(dv[0]);
            w.y = (unsigned)__double2hiint(dv[0]);
            w.z = (unsigned)__double2loint(dv[1]);
            w.w = (unsigned)__double2hiint(dv[1]);
            __builtin_amdgcn_raw_buffer_store_b128(w, drsrc, p * (DM * 8), parity_off, AUX_SC1);
            if (DM == 3) {
                u32x2 w2;
                w2.x = (unsigned)__double2loint(dv[DM - 1]);
                w2.y = (unsigned)__double2hiint(dv[DM - 1]);
                __builtin_amdgcn_raw_buffer_store_b64(w2, drsrc, p * (DM * 8) + 16, parity_off, AUX_SC1);
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < DM; ++cc) {
                u32x2 w2;
                w2.x = (unsigned)__double2loint(dv[cc]);
                w2.y = (unsigned)__double2hiint(dv[cc]);
                __builtin_amdgcn_raw_buffer_store_b64(w2, drsrc, (p * DM + cc) * 8, parity_off, AUX_SC1);
            }
        }
    };
    auto arm_d = [&](int32_t p, int32_t buf_off) {                // INB: the lane's words of a buffer <- sentinel
        u32x4 w;
        w.x = SENT_LO; w.y = SENT_HI; w.z = SENT_LO; w.w = SENT_HI;
        __builtin_amdgcn_raw_buffer_store_b128(w, drsrc, p * (DM * 8), buf_off, AUX_SC1);
        if (DM == 3) {
            u32x2 w2;
            w2.x = SENT_LO; w2.y = SENT_HI;
            __builtin_amdgcn_raw_buffer_store_b64(w2, drsrc, p * (DM * 8) + 16, buf_off, AUX_SC1);
        }
    };
    auto stale_d = [&](const double (&xv)[DM]) -> bool {
        bool st = false;
#pragma unroll
        for (int cc = 0; cc < DM; ++cc) st = st || ((unsigned)__double2hiint(xv[cc]) == SENT_HI);
        return st;
    };
    // nb (<= CH) consecutive block rows of a slice: columns and values (rows beyond nb: the column of the last one,
    // so that its gather stays in range; no values)
    // keep0 = first block row (of the slice) from which the loads are non-temporal (VAR & 1); rows below it keep the
    // default policy
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    auto load_rows = [&](const int32_t* __restrict__ bc_, const double2* __restrict__ vp, const double* __restrict__ vs,
                         int32_t j, int nb, int32_t keep0, int32_t (&col)[CH], double (&e)[CH][DD]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) col[u] = bc_[(int64_t)(j + max(0, min(u, nb - 1))) * 64];
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (u < nb) {
                if (NT && j + u >= keep0) {                                 // wave-uniform
#pragma unroll
                    for (int kp = 0; kp < NP; ++kp) {
                        const nt_d2 v2 = __builtin_nontemporal_load(
                            reinterpret_cast<const nt_d2*>(&vp[(int64_t)(j + u) * (DD * 32) + kp * 64]));
                        e[u][2 * kp] = v2.x;
                        e[u][2 * kp + 1] = v2.y;
                    }
                    if (DD & 1) e[u][DD - 1] = __builtin_nontemporal_load(&vs[(int64_t)(j + u) * (DD * 64)]);
                } else {
#pragma unroll
                    for (int kp = 0; kp < NP; ++kp) {
                        const double2 v2 = vp[(int64_t)(j + u) * (DD * 32) + kp * 64];
                        e[u][2 * kp] = v2.x;
                        e[u][2 * kp + 1] = v2.y;
                    }
                    if (DD & 1) e[u][DD - 1] = vs[(int64_t)(j + u) * (DD * 64)];
                }
            } else {
                // defined on every path: a conditionally rewritten loop-carried buffer would keep its old value alive
                // through the whole iteration (the prefetch buffer then costs 76 registers at the product's peak)
#pragma unroll
                for (int k = 0; k < DD; ++k) e[u][k] = 0.0;
            }
    };
    const int32_t* __restrict__ bc0 = a.bcol + OFFS(0) * 64 + lane;
    const double2* __restrict__ vp0 = reinterpret_cast<const double2*>(a.vals + OFFS(0) * (int64_t)(DD * 64)) + lane;
    const double* __restrict__ vs0 = a.vals + OFFS(0) * (int64_t)(DD * 64) + NP * 128 + lane;
    // rows of the batch prefetched during the synchronisation windows.  bit 4 of dbg: no prefetch (a layout variant, not
    // work skipping).  Tagged-granule form: wave 0 sweeps the granules during those windows, and VMEM returns in order
    // -- a prefetch of its own would sit in front of every sweep -- so wave 0 does not prefetch (the host hands it
    // one batch less of streamed rows instead)
#ifndef FEMCY_INB_PREFETCH
#define FEMCY_INB_PREFETCH 1
#endif
    // (seven slices of 3 x 3 blocks per wave: 210 registers of vectors -- no prefetch buffer, its 76 registers are what
    // the shape does not have)
    constexpr bool PREFETCH = !(DM == 3 && SPW >= 7);
    const int npf = (!PREFETCH || (a.dbg & 16) || (A2A && wave == 0) || (INB && !FEMCY_INB_PREFETCH)) ? 0 : max(0, min(CH, Ls[0] - JE(0)));
    int32_t pcol[CH];
    double pe[CH][DD];
    const int32_t jpf = npf > 0 ? JE(0) : 0;                               // (npf = 0: row 0's column, unused)
    if (PREFETCH) load_rows(bc0, vp0, vs0, jpf, npf, JE(0) + a.l2_rows, pcol, pe);
    // ---- x0 = 0, r = b, d = M r
    double xo[SPW][DM], rr[SPW][DM], mm[SPW][DM], dd[SPW][DM], Ad[SPW][DM];
    double accs = 0.0, accm = 0.0;
    uint32_t ownbits = 0;                                         // bit t * DM + c: this rank counts the DOF (MULTI)
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
#pragma unroll
        for (int c = 0; c < DM; ++c) {
            const bool in = node[t] >= 0;
            const int64_t i = in ? (int64_t)node[t] * DM + c : 0;
            // UNCONDITIONAL loads + selects (a padding lane reads entry 0): as `in ? a.b[i] : 0.0` each load sat in its own
            // exec-masked region, and in the 6- / 7-slice shapes of 3 x 3 blocks hipcc 7.0 placed register copies of
            // long-lived values (VGPR -> accumulation register) at the join of such a region BEFORE exec was restored --
            // the padding lanes of those registers kept garbage and the final store of x went through it
            // (profiles/r06_persist_spw67_fault.txt: a memory fault, or wrong iterates; -O1 was correct)
            const double bl = a.b[i], ml = a.M[i];
            const double bi = in ? bl : 0.0, mi = in ? ml : 0.0;
            xo[t][c] = 0.0;
            rr[t][c] = bi;
            mm[t][c] = mi;
            dd[t][c] = mi * bi;
            Ad[t][c] = 0.0;
            // across ranks a shared DOF is counted by its owner only; max|r| is the same on every replica
            const uint8_t own = MULTI ? a.owner[i] : (uint8_t)1;  // unconditional as well (see above)
            const bool mine = !MULTI || (in && own != 0);
            if (mine) ownbits |= 1u << (t * DM + c);
            if (mine) accs += bi * mi * bi;
            accm = fmax(accm, pabs(bi));
        }
        if (sl[t] >= 0 && (WIDE || node[t] >= 0)) publish_d(dpos[t], 0, dd[t]);
        if (INB && sl[t] >= 0) {                                  // buffers 1 and 2 start armed (drained with the first exchange)
            arm_d(dpos[t], a.npad * 8);
            arm_d(dpos[t], 2 * a.npad * 8);
        }
    }
    // across ranks: the lane's interface table entries (send entry, recv entry, neighbour | lower << 8, its nb_total) stay
    // in registers for the whole solve (round 5: they were re-loaded twice per iteration and used at once -- two exposed
    // L2 round trips on the critical path of every exchange); a wave without an interface lane skips both blocks
    // (the four-slice shape has no registers to spare: it keeps the wave-uniform flag and re-loads the entries)
    constexpr bool TABREG = SPW <= 3;
    int4 mtab[SPW];
    bool wave_has_if = false;
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
        mtab[t] = make_int4(-1, -1, 0, 0);
        if (MULTI && sl[t] >= 0) mtab[t] = reinterpret_cast<const int4*>(a.mr_tab)[(int64_t)sl[t] * 64 + lane];
        if (MULTI) wave_has_if = wave_has_if || __any(mtab[t].x >= 0 || mtab[t].y >= 0);
    }
    auto iface_entry = [&](int t) -> int4 {
        if (TABREG) return mtab[t];
        return sl[t] >= 0 ? reinterpret_cast<const int4*>(a.mr_tab)[(int64_t)sl[t] * 64 + lane] : make_int4(-1, -1, 0, 0);
    };
    unsigned round = 0;
    // across ranks: the cross-rank stage of an exchange (wave 0 of every workgroup; the local result is in `val`),
    // result through LDS to every thread
    int xr_it = -1;                                               // iteration the running exchanges belong to (-1: set-up)
    auto xrank_stage = [&](auto nv_tag, int base, double (&val)[decltype(nv_tag)::value],
                           const int (&op)[decltype(nv_tag)::value]) -> bool {
        constexpr int NV = decltype(nv_tag)::value;
        __syncthreads();
        if (wave == 0) {
            const bool okx = xrank_reduce<NV>(a, base, (xr_it + 1) & 1, mb_tag(a.tagbase, xr_it), val, op);
            if (lane == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v) bc[v] = val[v];
                if (!okx) s_fail = 1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) val[v] = bc[v];
        return s_fail == 0;
    };
    // granule arrays: [0, G) d.Ad, [G, 3 G) (r.M.r, max|r|), [3 G, 4 G) "d published"
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.slots, 0, G * 4 * 16, 0x00020000);
    auto poison_granules = [&]() {
        if (lane == 0) {
            granule_store(srsrc, blockIdx.x, 0.0, TAG_POISON);
            granule_store(srsrc, G + 2 * blockIdx.x, 0.0, TAG_POISON);
            granule_store(srsrc, G + 2 * blockIdx.x + 1, 0.0, TAG_POISON);
            granule_store(srsrc, 3 * G + blockIdx.x, 0.0, TAG_POISON);
        }
    };
    // (sum, max) of one pair per workgroup over the grid; the result is in every thread on return; false = time-out /
    // poison
    auto exchange_pair = [&](double s, double m, double* part2_base, double& s_out, double& m_out) -> bool {
        s = wave_sum(s);
        m = wave_max(m);
        if (lane == 0) {
            sm1[wave] = s;
            sm2[wave] = m;
        }
        __syncthreads();
        if (A2A) {
            if (wave == 0) {
                const double ws = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
                const double wm = fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3]));
                const unsigned long long tag = round + 1;
                if (lane == 0) granule_store(srsrc, G + 2 * blockIdx.x, ws, tag);
                if (lane == 1) granule_store(srsrc, G + 2 * blockIdx.x + 1, wm, tag);
                double o[2];
                const int op[2] = {0, 1};
                const bool okx = granule_sweep<2>(srsrc, G, G, tag, a.spin_limit, o, op);
                if (!okx) poison_granules();
                if (lane == 0) {
                    bc[0] = o[0];
                    bc[1] = o[1];
                    if (!okx) s_fail = 1;
                }
            }
            ++round;
            __syncthreads();
            s_out = bc[0];
            m_out = bc[1];
            return s_fail == 0;
        } else {
            if (tid == 0) {
                pst(part2_base + 2 * blockIdx.x, (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]));
                pst(part2_base + 2 * blockIdx.x + 1, fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3])));
            }
            if (!grid_barrier(a, round++, &s_fail)) return false;
            double ps = 0.0, pm = 0.0;
            for (int k = tid; k < G; k += PBS) {                                 // every workgroup: the same order
                ps += pld(part2_base + 2 * k);
                pm = fmax(pm, pld(part2_base + 2 * k + 1));
            }
            ps = wave_sum(ps);
            pm = wave_max(pm);
            __syncthreads();                                                     // sm1 / sm2 free again
            if (lane == 0) {
                sm1[wave] = ps;
                sm2[wave] = pm;
            }
            __syncthreads();
            s_out = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
            m_out = fmax(fmax(sm2[0], sm2[1]), fmax(sm2[2], sm2[3]));
            return true;
        }
    };
    auto exchange_pair_all = [&](double s_, double m_, double* part2_base, double& s_out, double& m_out) -> bool {
        if (!exchange_pair(s_, m_, part2_base, s_out, m_out)) return false;
        if (MULTI) {
            double val[2] = {s_out, m_out};
            const int op[2] = {0, 1};
            if (!xrank_stage(std::integral_constant<int, 2>{}, mb_sb(a.nranks), val, op)) return false;
            s_out = val[0];
            m_out = val[1];
        }
        return true;
    };
    double rMr = 0.0, r0 = 0.0;
    // the initial d is published with the first exchange: its stores are drained before anybody passes it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool ok0 = exchange_pair_all(accs, accm, a.part2, rMr, r0);
    double rmax = r0;
    int done = !ok0 ? 3 : ((r0 == 0.0) ? 1 : ((r0 != r0 || isinf(r0)) ? 2 : 0));
    int it = 0;

    // ---- Ad = K d for the wave's rows (d gathered with sc1 loads: the other XCDs wrote it with sc1 stores); returns
    // the lane's part of d.Ad
    bool pfail = false;                                           // V_INBAND: a gather never saw its value (time-out)
    auto product = [&](const int32_t poff) -> double {
        double acc[SPW][DM];
#pragma unroll
        for (int t = 0; t < SPW; ++t)
#pragma unroll
            for (int r = 0; r < DM; ++r) acc[t][r] = 0.0;
#pragma unroll
        for (int t = 0; t < SPW; ++t) {
            const int32_t L = Ls[t];
            const int32_t* __restrict__ bcp = a.bcol + OFFS(t) * 64 + lane;
            const double2* __restrict__ vp = reinterpret_cast<const double2*>(a.vals + OFFS(t) * (int64_t)(DD * 64)) + lane;
            const double* __restrict__ vs = a.vals + OFFS(t) * (int64_t)(DD * 64) + NP * 128 + lane;
            int32_t j = JE(t);                                               // first streamed block row
            // slice 0: its first streamed batch was loaded while the wave sat in the last synchronisation points
            if (PREFETCH && t == 0 && npf > 0 && !PDBG(a, 1)) {
                double xg[CH][DM];
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(pcol[u], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
                FEMCY_SETTLE(CH, _Pragma("unroll") for (int u = 0; u < CH; ++u) gather_d(pcol[u], poff, xg[u]))
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < npf) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc) acc[t][r] += pe[u][r * DM + cc] * xg[u][cc];
                    }
                j += npf;
            }
            // block rows held in registers: their gathers are issued in batches before the first multiply
            if (RJ > 0 && !PDBG(a, 4)) {
                constexpr int RB = RJ > 4 ? 3 : (RJ > 0 ? RJ : 1);               // rows per batch (register budget)
#pragma unroll
                for (int j0 = 0; j0 < RJ; j0 += RB) {
                    double xg[RB][DM];
                    // (OWN_DIAG: block row 0 is the diagonal block -- its column is the lane's own d, which is at hand)
#define FEMCY_REG_GATHER                                                                       \
    _Pragma("unroll") for (int u = 0; u < RB; ++u) {                                          \
        if (OWN_DIAG && j0 + u == 0) {                                                         \
            _Pragma("unroll") for (int cc = 0; cc < DM; ++cc) xg[u][cc] = dd[t][cc];          \
        } else if (j0 + u < RJ) {                                                              \
            gather_d(rcl[t][j0 + u], poff, xg[u]);                                             \
        } else {                                                                               \
            _Pragma("unroll") for (int cc = 0; cc < DM; ++cc) xg[u][cc] = 0.0;                \
        }                                                                                      \
    }
                    FEMCY_REG_GATHER
                    __builtin_amdgcn_sched_barrier(0);
                    FEMCY_SETTLE(RB, FEMCY_REG_GATHER)
#undef FEMCY_REG_GATHER
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        if (j0 + u < RJ) {
#pragma unroll
                            for (int r = 0; r < DM; ++r)
#pragma unroll
                                for (int cc = 0; cc < DM; ++cc) acc[t][r] += rv[t][j0 + u][r * DM + cc] * xg[u][cc];
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // block rows held in LDS, CH at a time (a short batch repeats its last row's gather and skips the multiply)
            for (int32_t jr = JL(t); jr < JE(t) && !PDBG(a, 2); jr += CH) {
                const int nb = min(CH, JE(t) - jr);
                const int q = QL(t) + (jr - JL(t));
                double xg[CH][DM];
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(lcols[(q + min(u, nb - 1)) * 64 + lane], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
                FEMCY_SETTLE(CH, _Pragma("unroll") for (int u = 0; u < CH; ++u) gather_d(lcols[(q + min(u, nb - 1)) * 64 + lane], poff, xg[u]))
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < nb) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc)
                                acc[t][r] += lvals[((q + u) * DD + r * DM + cc) * 64 + lane] * xg[u][cc];
                    }
            }
#ifdef FEMCY_PERSIST_PIPE
            // (round 5 experiment, not in the shipped build) software-pipelined stream: the loads of batch k + 1 are
            // issued behind the gathers of batch k and travel while batch k is multiplied -- two batches of values in
            // registers (A / B ping-pong), for matrices that stream from HBM rather than from the Infinity Cache
            if (j < L && !PDBG(a, 1)) {
                int32_t colA[CH], colB[CH];
                double eA[CH][DD], eB[CH][DD];
                int nbA = min(CH, L - j), nbB = 0;
                load_rows(bcp, vp, vs, j, nbA, JE(t) + a.l2_rows, colA, eA);
                for (;;) {
                    {
                        double xg[CH][DM];
#pragma unroll
                        for (int u = 0; u < CH; ++u) gather_d(colA[u], poff, xg[u]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int32_t jn = j + nbA;
                        nbB = max(0, min(CH, L - jn));
                        load_rows(bcp, vp, vs, nbB > 0 ? jn : j, nbB, JE(t) + a.l2_rows, colB, eB);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                            if (u < nbA) {
#pragma unroll
                                for (int r = 0; r < DM; ++r)
#pragma unroll
                                    for (int cc = 0; cc < DM; ++cc) acc[t][r] += eA[u][r * DM + cc] * xg[u][cc];
                            }
                        j = jn;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (nbB == 0) break;
                    {
                        double xg[CH][DM];
#pragma unroll
                        for (int u = 0; u < CH; ++u) gather_d(colB[u], poff, xg[u]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int32_t jn = j + nbB;
                        nbA = max(0, min(CH, L - jn));
                        load_rows(bcp, vp, vs, nbA > 0 ? jn : j, nbA, JE(t) + a.l2_rows, colA, eA);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                            if (u < nbB) {
#pragma unroll
                                for (int r = 0; r < DM; ++r)
#pragma unroll
                                    for (int cc = 0; cc < DM; ++cc) acc[t][r] += eB[u][r * DM + cc] * xg[u][cc];
                            }
                        j = jn;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (nbA == 0) break;
                }
            }
#endif
            // streamed block rows, CH at a time: columns, then the values, then the gathers, then the multiplies
            while (j < L && !PDBG(a, 1)) {
                const int nb = min(CH, L - j);
                int32_t col[CH];
                double e[CH][DD], xg[CH][DM];
                load_rows(bcp, vp, vs, j, nb, JE(t) + a.l2_rows, col, e);
#pragma unroll
                for (int u = 0; u < CH; ++u) gather_d(col[u], poff, xg[u]);
                __builtin_amdgcn_sched_barrier(0);
                FEMCY_SETTLE(CH, _Pragma("unroll") for (int u = 0; u < CH; ++u) gather_d(col[u], poff, xg[u]))
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (u < nb) {
#pragma unroll
                        for (int r = 0; r < DM; ++r)
#pragma unroll
                            for (int cc = 0; cc < DM; ++cc) acc[t][r] += e[u][r * DM + cc] * xg[u][cc];
                    }
                j += nb;
            }
        }
        double dot = 0.0;
#pragma unroll
        for (int t = 0; t < SPW; ++t)
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                Ad[t][r] = acc[t][r];
                if (IN(t)) dot += dd[t][r] * acc[t][r];
            }
        return dot;
    };

    auto iteration = [&]() {
        __syncthreads();                                                     // sm1 / sm2 of the previous phase are read
        const int32_t poff = (INB ? it % 3 : (it & 1)) * a.npad * 8;          // byte offset of this iteration's d
        double dot = product(poff);
        if (INB && pfail && lane == 0) s_fail = 1;                           // seen by everybody behind the next barrier
        xr_it = it;
        if (MULTI) {
            // interface rows: the partial sums of this rank go straight into the sharing rank's mailbox (system-scope
            // 8-byte words that validate themselves); they travel while the exchanges below run
            const uint32_t tag = mb_tag(a.tagbase, it);
            if (wave_has_if) {
#pragma unroll
                for (int t = 0; t < SPW; ++t) {
                    const int4 tb = iface_entry(t);
                    if (tb.x >= 0) {
                        unsigned long long* dst = a.peer[tb.z & 0xff] + mb_ad(a.nranks) +
                                                  ((size_t)(it & 1) * tb.w + tb.x) * 2;
#pragma unroll
                        for (int c = 0; c < DM; ++c) mb_store(dst + 2 * c, Ad[t][c], tag);
                    }
                }
            }
        }
        // the matrix does not change: slice 0's first streamed batch for the NEXT product is requested inside the first
        // exchange (after the arrival, so that it does not delay it) and arrives while the wave waits in the three
        // synchronisation points (registers and memory system are idle there)
        dot = wave_sum(dot);
        if (lane == 0) sm1[wave] = dot;
        __syncthreads();
        double dAd;
        if (INB && s_fail) {                                                 // a gather timed out in this workgroup:
            if (wave == 0) poison_granules();                                // release everybody with the same verdict
            done = 3;
            return;
        }
        if (A2A) {
            if (wave == 0) {
                const double ws = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
                const unsigned long long tag = round + 1;
                if (lane == 0) granule_store(srsrc, blockIdx.x, ws, tag);
                double o[1];
                const int op[1] = {0};
                const bool okx = granule_sweep<1>(srsrc, 0, G, tag, a.spin_limit, o, op);
                if (!okx) poison_granules();
                if (lane == 0) {
                    bc[0] = o[0];
                    if (!okx) s_fail = 1;
                }
                // defined on every path (see load_rows): wave 0 holds no prefetched batch
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    pcol[u] = 0;
#pragma unroll
                    for (int k = 0; k < DD; ++k) pe[u][k] = 0.0;
                }
            } else if (PREFETCH) {
                load_rows(bc0, vp0, vs0, jpf, npf, JE(0) + a.l2_rows, pcol, pe);
            }
            ++round;
            __syncthreads();
            if (s_fail) { done = 3; return; }
            dAd = bc[0];
        } else {
            if (tid == 0) pst(a.part1 + (size_t)(it & 1) * G + blockIdx.x, (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]));
            if (!grid_barrier(a, round++, &s_fail, [&] { if (PREFETCH) load_rows(bc0, vp0, vs0, jpf, npf, JE(0) + a.l2_rows, pcol, pe); })) { done = 3; return; }
            double ps = 0.0;
            for (int k = tid; k < G; k += PBS) ps += pld(a.part1 + (size_t)(it & 1) * G + k);
            ps = wave_sum(ps);
            __syncthreads();
            if (lane == 0) sm1[wave] = ps;
            __syncthreads();
            dAd = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
        }
        if (MULTI) {
            double val[1] = {dAd};
            const int op[1] = {0};
            if (!xrank_stage(std::integral_constant<int, 1>{}, MB_SA, val, op)) { done = 3; return; }
            dAd = val[0];
            // the neighbour's partial sums of the interface rows, added in ascending rank order on both sides
            const uint32_t tag = mb_tag(a.tagbase, it);
            bool okr = true;
            if (wave_has_if) {
#pragma unroll
                for (int t = 0; t < SPW; ++t) {
                    const int4 tb = iface_entry(t);
                    if (tb.y >= 0) {
                        const unsigned long long* src = a.mbox + mb_ad(a.nranks) + ((size_t)(it & 1) * a.nb_total + tb.y) * 2;
#pragma unroll
                        for (int c = 0; c < DM; ++c) {
                            double other = 0.0;
                            uint32_t spins = 0;
                            while (!mb_load(src + 2 * c, tag, other)) {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > a.xspin_limit) { okr = false; break; }
                            }
                            Ad[t][c] = (tb.z & 0x100) ? other + Ad[t][c] : Ad[t][c] + other;
                        }
                    }
                }
            }
            if (__any(!okr)) s_fail = 1;
            __syncthreads();
            if (s_fail) { done = 3; return; }
        }
        // ---- alpha; x, r; partials of (r.M.r, max|r|)
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
                if (IN(t)) {
                    if (!MULTI || ((ownbits >> (t * DM + c)) & 1u)) accs += ri * mm[t][c] * ri;
                    accm = fmax(accm, pabs(ri));
                }
            }
        __syncthreads();
        double rMr_new = 0.0;
        if (!exchange_pair_all(accs, accm, a.part2 + (size_t)((it + 1) & 1) * 2 * G, rMr_new, rmax)) { done = 3; return; }
        ++it;
        if (PDBG(a, 15)) rmax = 1.0, rMr_new = 1.0;   // bits 0-3 skip work: keep iterating on whatever numbers result
        if (rmax != rmax || isinf(rmax) || rMr_new != rMr_new) {
            done = 2;
        } else if (rmax < a.eps * r0) {
            done = 1;
        } else {
            // ---- d = M r + beta d, published for the next product
            const double beta = rMr_new / rMr;
            const int32_t noff = (INB ? it % 3 : (it & 1)) * a.npad * 8;
#pragma unroll
            for (int t = 0; t < SPW; ++t) {
#pragma unroll
                for (int c = 0; c < DM; ++c) dd[t][c] = mm[t][c] * rr[t][c] + beta * dd[t][c];
                if (sl[t] >= 0 && (WIDE || IN(t))) publish_d(dpos[t], noff, dd[t]);
                // V_INBAND: re-arm the buffer the iteration after next publishes into (last read one iteration ago)
                if (INB && sl[t] >= 0) arm_d(dpos[t], ((it + 1) % 3) * a.npad * 8);
            }
            rMr = rMr_new;
            if (!INB && it < a.maxit) {
                if (A2A) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's part of d has left
                    __syncthreads();
                    if (wave == 0) {
                        const unsigned long long tag = round + 1;
                        if (lane == 0) granule_store(srsrc, 3 * G + blockIdx.x, 0.0, tag);
                        double o[1];
                        const int op[1] = {0};
                        const bool okx = granule_sweep<1>(srsrc, 3 * G, G, tag, a.spin_limit, o, op);
                        if (!okx) poison_granules();
                        if (lane == 0 && !okx) s_fail = 1;
                    }
                    ++round;
                    __syncthreads();
                    if (s_fail) { done = 3; return; }
                } else if (!grid_barrier(a, round++, &s_fail)) {
                    done = 3;
                    return;
                }
            }
        }
        if (done) rMr = rMr_new;
    };
    while (!done && it < a.maxit) iteration();
#pragma unroll
    for (int t = 0; t < SPW; ++t)
        if (IN(t)) {
            const int64_t nd = a.node_of[(int64_t)sl[t] * 64 + lane];
#pragma unroll
            for (int c = 0; c < DM; ++c) a.x[nd * DM + c] = xo[t][c];
        }
    if (blockIdx.x == 0 && tid == 0) {
        a.st->iters = it;
        a.st->r0 = r0;
        a.st->rmax = rmax;
        a.st->done = done;
        a.st->rMr[0] = rMr;
    }
}
#undef IN
#undef OFFS
#undef JL
#undef QL
#undef JE

// ------------------------------------------------------------------------------------------------ ceiling probes
// What the persistent kernel runs against (bench.py's roofline): (1) the rate at which the chip streams a read-only
// buffer of the size of the kernel's streamed matrix part, in the kernel's launch shape (one workgroup of four waves
// per CU, 16-byte loads, XCD-contiguous ranges), repeated so that it comes from wherever a buffer of that size lives
// (Infinity Cache below ~200 MB); (2) the price of one grid-wide exchange of the form the solver uses.
template <int U>   // 16-byte loads in flight per lane
__global__ void __launch_bounds__(PBS) k_probe_stream(const double2* __restrict__ buf, int64_t n16, int reps, int nt,
                                                      double* __restrict__ sink) {
    const int G = gridDim.x, xk = blockIdx.x % PNX, per = G / PNX;
    const int64_t chunk = (n16 / G) / (PBS * U) * (PBS * U);     // 16-byte elements per workgroup, whole tiles
    double s0 = 0.0, s1 = 0.0;
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    for (int r = 0; r < reps; ++r) {
        // XCD k walks a contiguous range; the chunk of a workgroup moves on by a third of that range every pass, so that
        // nothing is re-read before the whole buffer has gone by (a workgroup looping over ONE small chunk would
        // measure its L2, not the stream: 24 TB/s at 8 workgroups per CU in the first version of this probe)
        const int64_t wg = (int64_t)xk * per + (blockIdx.x / PNX + (int64_t)r * (per / 3 + 1)) % per;
        const double2* __restrict__ p = buf + wg * chunk + threadIdx.x;
        for (int64_t i = 0; i < chunk; i += PBS * U) {
            double2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (nt) {
                    const nt_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p + i + u * PBS));
                    v[u] = make_double2(t.x, t.y);
                } else {
                    v[u] = p[i + u * PBS];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s0 += v[u].x;
                s1 += v[u].y;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (s0 + s1 == 1.2345e-300) sink[0] = s0;                    // keeps the loads alive
}

// the same sweep through the LDS-DMA path (global_load_lds_dwordx4: 1 KiB per wave instruction lands in LDS without
// passing through VGPRs), U instructions in flight per wave into a wave-private ring of U KiB; one ds_read per lane and
// tile keeps the data "used"
template <int U>
__global__ void __launch_bounds__(PBS) k_probe_stream_lds(const double2* __restrict__ buf, int64_t n16, int reps, int nt,
                                                          double* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds_probe_ring[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int G = gridDim.x, xk = blockIdx.x % PNX, per = G / PNX;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t chunk = (n16 / G) / (PBS * U) * (PBS * U);
    char* ring = lds_probe_ring + (size_t)wave * (U * 1024);
    double s0 = 0.0;
    for (int r = 0; r < reps; ++r) {
        const int64_t wg = (int64_t)xk * per + (blockIdx.x / PNX + (int64_t)r * (per / 3 + 1)) % per;
        const double2* __restrict__ p = buf + wg * chunk + (int64_t)wave * (chunk / 4) + lane;
        for (int64_t i = 0; i < chunk / 4; i += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (nt) __builtin_amdgcn_global_load_lds(p + i + u * 64, (lds_ptr_t)(ring + u * 1024), 16, 0, 2);
                else __builtin_amdgcn_global_load_lds(p + i + u * 64, (lds_ptr_t)(ring + u * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s0 += reinterpret_cast<const double*>(ring)[lane * 2 + (int)(i & 1)];
        }
    }
    if (s0 == 1.2345e-300) sink[0] = s0;
}

template <int A2A>
__global__ void __launch_bounds__(PBS) k_probe_exchange(PersistPcg a, int rounds, double* __restrict__ out) {
    __shared__ double sm1[PBS / 64], bc[2];
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = gridDim.x;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.slots, 0, G * 4 * 16, 0x00020000);
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const double mine = (double)(blockIdx.x + r);
        double total;
        if (A2A) {
            // three granule arrays in turn, as in the solver: a slot is rewritten only after two other exchanges
            const int base = (r % 3) * G;
            if (wave == 0) {
                if (lane == 0) granule_store(srsrc, base + blockIdx.x, mine, (unsigned long long)r + 1);
                double o[1];
                const int op[1] = {0};
                const bool okx = granule_sweep<1>(srsrc, base, G, (unsigned long long)r + 1, a.spin_limit, o, op);
                if (lane == 0) {
                    bc[0] = o[0];
                    if (!okx) s_fail = 1;
                }
            }
            __syncthreads();
            total = bc[0];
            __syncthreads();
        } else {
            if (tid == 0) pst(a.part1 + (size_t)(r & 1) * G + blockIdx.x, mine);
            if (!grid_barrier(a, (unsigned)r, &s_fail)) break;
            double ps = 0.0;
            for (int k = tid; k < G; k += PBS) ps += pld(a.part1 + (size_t)(r & 1) * G + k);
            ps = wave_sum(ps);
            __syncthreads();
            if (lane == 0) sm1[wave] = ps;
            __syncthreads();
            total = (sm1[0] + sm1[1]) + (sm1[2] + sm1[3]);
        }
        if (s_fail) break;
        acc += total;
    }
    if (blockIdx.x == 0 && tid == 0) {
        out[0] = acc;
        out[1] = s_fail ? -1.0 : 0.0;
    }
}

// one cross-rank reduction of the persistent multi-rank PCG, repeated: a single wave per rank writes its value into
// every rank's mailbox and polls its own (xrank_reduce, the code the solver runs) -- the mailbox round trip between
// the ranks' kernels, link latency and skew included.  out[0] = sum of the reduced values, out[1] = -1 on a time-out
// out[2] = ticks of the constant-rate wall clock (s_memrealtime) spent in rounds 1 .. rounds - 1 (round 0 absorbs the
// skew between the ranks' launches)
__global__ void __launch_bounds__(64) k_probe_mailbox(PersistPcg a, int rounds, double* __restrict__ out) {
    double acc = 0.0;
    bool ok = true;
    unsigned long long t0 = 0;
    for (int r = 0; r < rounds && ok; ++r) {
        if (r == 1) t0 = wall_clock64();
        double val[1] = {(double)(a.rank + 1)};
        const int op[1] = {0};
        ok = xrank_reduce<1>(a, MB_SA, r & 1, mb_tag(a.tagbase, r), val, op);
        acc += val[0];
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = acc;
        out[1] = ok ? 0.0 : -1.0;
        out[2] = (double)(t1 - t0);
    }
}

int persist_buffers(Ctx* c, int G, int64_t npad, PersistPcg* a) {
    const int64_t need = 3 * npad + 2 * G + 4 * G + 8 * G + 160;  // d (<= 3 buffers) + partials + granules (4 G x 16 B) + counters
    if (!c->d_persist || c->persist_cap < need) {
        if (c->d_persist) (void)hipFree(c->d_persist);
        c->d_persist = nullptr;
        c->persist_cap = need;
        FEMCY_HIP(dmalloc(&c->d_persist, sizeof(double) * need));
    }
    a->dbuf = c->d_persist;
    a->part1 = c->d_persist + 3 * npad;
    a->part2 = a->part1 + 2 * G;
    a->slots = a->part2 + 4 * G;
    a->xc = reinterpret_cast<unsigned int*>(a->slots + 8 * G);
    a->top = a->xc + 8 * 32;
    a->spin_limit = c->barrier_spin_limit;
    a->xspin_limit = (uint32_t)std::min<uint64_t>((uint64_t)c->barrier_spin_limit * 16, 1u << 30);   // ~10 s
    a->dbg = c->opt_persist_dbg;
    // granules and counters start at zero (tags / rounds count from 1)
    FEMCY_HIP(hipMemsetAsync(a->slots, 0, sizeof(double) * 8 * G + sizeof(unsigned int) * (8 * 32 + 32), c->stream));
    return FEMCY_OK;
}

}  // namespace

// block columns as storage positions (VAR & 4): bcolp[i] = pos[bcol[i]]
__global__ void k_bcol_to_pos(int64_t n, const int32_t* __restrict__ bcol, const int32_t* __restrict__ pos,
                              int32_t* __restrict__ bcolp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) bcolp[i] = pos[bcol[i]];
}
int ensure_bcolp(Ctx* c) {
    if (c->bcolp_serial == c->pattern_serial && c->d_bcolp) return FEMCY_OK;
    if (c->d_bcolp) (void)hipFree(c->d_bcolp);
    c->d_bcolp = nullptr;
    const int64_t nb = c->stored_rows * SLICE;
    FEMCY_HIP(dmalloc(&c->d_bcolp, std::max<int64_t>(nb, 1) * sizeof(int32_t)));
    hipLaunchKernelGGL(k_bcol_to_pos, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, nb,
                       (const int32_t*)c->d_bcol, (const int32_t*)c->d_pos, c->d_bcolp);
    FEMCY_HIP(hipGetLastError());
    c->bcolp_serial = c->pattern_serial;
    return FEMCY_OK;
}

// default variant of the persistent kernel (FEMCY_TUNE_PERSIST_VARIANT = -1), chosen by the round-3 measurements
// 
#ifndef FEMCY_PERSIST_DEFAULT_VARIANT
#define FEMCY_PERSIST_DEFAULT_VARIANT 6
#endif

constexpr int PERSIST_MULTI_RETRY = 16;   // RCCL-loop solves after a cross-rank time-out before the one-launch path is tried again

int64_t persist_streamed_bytes(Ctx* c);

// the launch shape of the persistent kernel for this pattern: slices per wave (SPW: the kernel's register arrays),
// block rows per slice in registers (RJ) and per wave in LDS.  3 x 3 blocks: SPW 3 (RJ 4 / 5) or 4 (RJ 3) -- up to
// 4 096 slices = 786 k DOF -- and, round 6, 5 / 6 (RJ 1) and 7 (RJ 0): up to 7 168 slices = 1.37 M DOF; 2 x 2 blocks (round 5): additionally SPW 6 (RJ 3) and 8 (RJ 2) -- a lane's vectors take
// 20 registers per slice instead of 30, a block row 9 instead of 19 -- up to 8 192 slices = 1.05 M DOF in 2-D
// (BASELINE configs[1] at the size of the 3-D headline system).  false = the pattern does not fit.
struct PersistShape { int G, nwx, SPW, rj, lds_rows; int32_t maxrange; };
bool persist_shape(const Ctx* c, PersistShape* out) {
    PersistShape sh{};
    sh.G = ((c->opt_persist_wgs > 0 ? c->opt_persist_wgs : c->persist_cus) / PNX) * PNX;   // one workgroup per CU
    if (sh.G < PNX || !c->have_pattern) return false;
    sh.nwx = (sh.G / PNX) * 4;
    sh.maxrange = 0;
    for (int k = 0; k < PNX; ++k) sh.maxrange = std::max(sh.maxrange, c->xcd.start[k + 1] - c->xcd.start[k]);
    const int per_wave = (sh.maxrange + sh.nwx - 1) / sh.nwx;
    const bool on = c->opt_persist_rj != 0;
    if (c->dm == 3) {
        if (per_wave > 7) return false;
        sh.SPW = per_wave > 3 ? per_wave : 3;
        if (const char* e = getenv("FEMCY_DEBUG_FORCE_SPW")) sh.SPW = std::min(7, std::max(sh.SPW, atoi(e)));
        sh.rj = sh.SPW == 3 ? c->opt_persist_rj : (!on ? 0 : (sh.SPW == 4 ? 3 : (sh.SPW == 7 ? 0 : 1)));
    } else {
        if (per_wave > 8) return false;
        sh.SPW = per_wave > 6 ? 8 : (per_wave > 4 ? 6 : (per_wave > 3 ? 4 : 3));
        sh.rj = !on ? 0 : (sh.SPW == 8 ? 2 : (sh.SPW == 6 ? 3 : 5));
    }
    const int DD = c->dm * c->dm;
    int lds_rows = c->opt_persist_lds < 0 ? (int)((c->small_max_lds - 2048) / (4 * 64 * (DD * 8 + 4))) : c->opt_persist_lds;
    sh.lds_rows = std::max(0, std::min(lds_rows, sh.SPW * (int)c->max_row_blocks));
    *out = sh;
    return true;
}

// does the system of this context take the persistent kernel?  (the multi-rank agreement)
bool persist_pattern_fits(Ctx* c) {
    if (!c->opt_persist || !c->have_pattern) return false;
    PersistShape sh;
    // (the multi-rank kernel is instantiated for 3 x 3 blocks with three or four slices per wave and -- round 6 -- for every
    // 2 x 2 shape)
    if (!persist_shape(c, &sh) || (c->dm == 3 && sh.SPW > 4)) return false;
    const int G = sh.G;
    if (c->opt_persist >= 2) return true;
    // evaluated here once so that every rank applies the same verdict.  The single-rank rule "the chip is filled 1.5 times over" (below ~380 slices three launches are
    // as fast) does not apply across ranks: what the one-launch path competes with there is three launches PLUS two
    // collectives per iteration (49 us against 32 at 1 M elements per rank; a strong-scaling slab of the 1 M plate on
    // 8 ranks has 375 slices)
    if (c->nslices < G / 2) return false;
    // ACROSS RANKS the streamed part still has to fit the Infinity Cache (240 MiB, the rule of rounds 2-4): round 5 lifted
    // the limit for a single rank on a measurement (124 k C3D10, 287 MB streamed from HBM: 61 against 78 us), but no
    // multi-rank run has streamed a larger matrix from HBM while polling its mailboxes -- the rule stays until one has
    // (FEMCY_TUNE_PERSIST_MAX_MB lowers it further; FEMCY_OPT_PCG_PERSIST = 2 above takes any size)
    const int64_t limit = std::min<int64_t>(c->persist_max_bytes, (int64_t)240 << 20);
    const int64_t row_bytes = (int64_t)(c->dm * c->dm * 8 + 4) * 64;
    return c->stored_rows * row_bytes <= limit || persist_streamed_bytes(c) <= limit;
}

// eligibility + launch; *handled = false when the system does not qualify (too small, too large, ranks not agreed)
int pcg_persist_solve(Ctx* c, const double* d_b, double* d_x, double eps, int32_t maxit, bool* handled) {
    *handled = false;
    const bool multi = c->comm != nullptr;
    if (!c->opt_persist || c->persist_failed) return FEMCY_OK;
    if (multi && c->persist_multi && c->persist_multi_failed && c->opt_persist_multi) {
        // a time-out is usually host-side skew between the ranks' calls (module load, a Python GC pause), not a broken
        // node: after PERSIST_MULTI_RETRY solves on the RCCL loop the one-launch path is tried again.  Every rank makes
        // the same sequence of femcy_pcg calls and learnt of the failure in the same collective, so the counters agree.
        if (++c->persist_multi_fallbacks >= PERSIST_MULTI_RETRY) {
            c->persist_multi_failed = false;
            c->persist_multi_fallbacks = 0;
        }
    }
    if (multi && (!c->persist_multi || c->persist_multi_failed || !c->opt_persist_multi)) return FEMCY_OK;
    PersistShape sh;
    if (!persist_shape(c, &sh)) return FEMCY_OK;
    const int G = sh.G, nwx = sh.nwx;
    // every wave gets its slices (<= 4; <= 8 in 2-D), and the chip is filled 1.5 times over: below ~380 slices the 13 us of
    // synchronisation per iteration exceed the (graph-replayed) three-launch iteration (size sweep in DESIGN.md)
    // (FEMCY_OPT_PCG_PERSIST = 2 takes any system whose slices fit; waves without a slice idle through the exchanges)
    // across ranks every rank must take the same path: only the rule the agreement checked applies there
    if ((c->nslices < G + G / 2) && c->opt_persist < 2 && !multi) return FEMCY_OK;
    const int DD = c->dm * c->dm;
    const int SPW = sh.SPW, lds_rows = sh.lds_rows;               // slices per wave (the kernel's register arrays)
    // ... the part of the matrix that is STREAMED every iteration may be limited (persist_max_bytes, ctx.hpp; no limit
    // by default since round 5).  Measured: 1.4 M C3D4 elements (277 MB stored, 100 MB of it resident) 41.7 us per
    // iteration here against 63.1 with three launches; 124 k C3D10 (380 MB stored, 287 MB streamed from HBM) 61.0 us
    // here against 78.0 (round 2, before the nt stream / tagged granules / storage-order d: 99-103 against 93).
    const int rj = sh.rj;
    const int64_t row_bytes = (int64_t)(DD * 8 + 4) * 64;
    const int64_t kbytes = c->stored_rows * row_bytes;
    const int64_t resident = (int64_t)G * 4 * (SPW * rj + lds_rows) * row_bytes;   // upper bound (short slices hold less)
    if (kbytes - resident > c->persist_max_bytes && c->opt_persist < 2 && !multi) return FEMCY_OK;
    const size_t lds = (size_t)4 * lds_rows * 64 * (DD * 8 + 4) + 16;
    const int var = c->opt_persist_variant < 0 ? FEMCY_PERSIST_DEFAULT_VARIANT : c->opt_persist_variant;
    const bool wide = (var & V_WIDE) != 0, a2a = (var & V_A2A) != 0;
    // d in storage order covers the padding lanes of the last slice as well
    const int64_t npad = wide ? (((int64_t)c->nslices * SLICE * c->dm + 1) & ~(int64_t)1) : ((c->n + 1) & ~(int64_t)1);
    // set-up and launch.  Across ranks every path from here on has to reach the agreement below -- the other ranks are
    // in it -- so a failure in this part is recorded, voted "bad", and returned only afterwards
    size_t tp = (size_t)-1;
    bool launched = true, not_resident = false;
    auto setup_and_launch = [&]() -> int {
        PersistPcg a;
        {
            int rc = persist_buffers(c, G, npad, &a);
            if (rc) return rc;
        }
        // slices of each wave: XCD k's waves share the slice range xcd[k] .. xcd[k+1] (the ranges are balanced by stored
        // block rows); inside it the slices go longest first to the wave with the least rows so far (LPT) -- the sigma-
        // sorted windows would otherwise hand all long slices to the same waves.  Tagged-granule form: wave 0 of every
        // workgroup sweeps the granules and does not prefetch, so it starts with a handicap of one batch
        {
            std::vector<int64_t> key = {G, SPW, c->pattern_serial, a2a ? 1 : 0};
            for (int k = 0; k <= PNX; ++k) key.push_back(c->xcd.start[k]);
            if (key != c->persist_assign_key || !c->d_persist_assign) {
                std::vector<int32_t> assign((size_t)PNX * nwx * SPW, -1);
                std::vector<int32_t> order, load(nwx), cnt(nwx);
                for (int k = 0; k < PNX; ++k) {
                    order.clear();
                    for (int32_t s = c->xcd.start[k]; s < c->xcd.start[k + 1]; ++s) order.push_back(s);
                    std::stable_sort(order.begin(), order.end(),
                                     [&](int32_t x, int32_t y) { return c->h_slice_len[x] > c->h_slice_len[y]; });
                    for (int w = 0; w < nwx; ++w) load[w] = (a2a && (w % 4) == 0) ? CH : 0;
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
                FEMCY_HIP(dmalloc(&c->d_persist_assign, assign.size() * sizeof(int32_t)));
                FEMCY_HIP(hipMemcpyAsync(c->d_persist_assign, assign.data(), assign.size() * sizeof(int32_t),
                                         hipMemcpyHostToDevice, c->stream));
                FEMCY_HIP(hipStreamSynchronize(c->stream));
                c->persist_assign_key = key;
            }
        }
        if (wide) {
            int rc = ensure_bcolp(c);
            if (rc) return rc;
        }
        a.assign = c->d_persist_assign;
        a.slice_len = c->d_slice_len; a.slice_off = c->d_slice_off; a.bcol = wide ? c->d_bcolp : c->d_bcol; a.node_of = c->d_node_of;
        a.vals = c->d_Kvals; a.b = d_b; a.M = c->d_M; a.x = d_x;
        a.st = c->d_state;
        a.npad = (int32_t)npad; a.maxit = maxit; a.lds_rows = lds_rows; a.eps = eps;
        a.l2_rows = c->opt_persist_l2rows;
        a.mbox = c->d_mbox; a.peer = c->d_peer_tab; a.mr_tab = c->d_mr_tab; a.owner = c->d_owner;
        a.rank = c->rank; a.nranks = c->nranks; a.nb_total = c->h_nb_ptr.empty() ? 0 : c->h_nb_ptr.back();
        c->solve_serial = (c->solve_serial % 4095) + 1;               // 1 .. 4095: a tag is never 0 (the zeroed mailbox)
        a.tagbase = c->solve_serial << 20;
#define FEMCY_PERSIST_M(DM_, SPW_, RJ_, VAR_, MULTI_)                                                             \
        do {                                                                                                          \
            const void* fn = reinterpret_cast<const void*>(&k_pcg_persist<DM_, SPW_, RJ_, VAR_, MULTI_>);             \
            if (lds > 48 * 1024)                                                                                      \
                FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
            /* the grid barrier needs all G workgroups resident; across ranks a refusal here would leave the others */ \
            /* polling, so it is reported as a failed solve: the agreement after the launch sends everybody to RCCL  */ \
            if (!coresident(c, fn, PBS, lds, G)) {                                                                    \
                not_resident = true;                                                                                  \
                launched = false;                                                                                     \
                break;                                                                                                \
            }                                                                                                         \
            tp = timing_begin(c, T_PERSIST);                                                                          \
            hipLaunchKernelGGL((k_pcg_persist<DM_, SPW_, RJ_, VAR_, MULTI_>), dim3(G), dim3(PBS), lds, c->stream, a); \
        } while (0)
#define FEMCY_PERSIST(DM_, SPW_, RJ_, VAR_) FEMCY_PERSIST_M(DM_, SPW_, RJ_, VAR_, false)
#ifdef FEMCY_PERSIST_ALL_VARIANTS
#define FEMCY_PERSIST_V(DM_, SPW_, RJ_)                                                                           \
        switch (var & 15) {                                                                                           \
            case 14: FEMCY_PERSIST(DM_, SPW_, RJ_, 14); break;                                                        \
            case 0: FEMCY_PERSIST(DM_, SPW_, RJ_, 0); break;                                                          \
            case 1: FEMCY_PERSIST(DM_, SPW_, RJ_, 1); break;                                                          \
            case 2: FEMCY_PERSIST(DM_, SPW_, RJ_, 2); break;                                                          \
            case 3: FEMCY_PERSIST(DM_, SPW_, RJ_, 3); break;                                                          \
            case 4: FEMCY_PERSIST(DM_, SPW_, RJ_, 4); break;                                                          \
            case 5: FEMCY_PERSIST(DM_, SPW_, RJ_, 5); break;                                                          \
            case 6: FEMCY_PERSIST(DM_, SPW_, RJ_, 6); break;                                                          \
            default: FEMCY_PERSIST(DM_, SPW_, RJ_, 7); break;                                                         \
        }
#else
        // the shipped library carries the default variant, the other one of {6 (three exchanges), 14 (two exchanges +
        // in-band d)} and the round-2 form (0) of every shape
#define FEMCY_PERSIST_ALT_VARIANT (FEMCY_PERSIST_DEFAULT_VARIANT == 14 ? 6 : 14)
#define FEMCY_PERSIST_V(DM_, SPW_, RJ_)                                                                           \
        if ((var & 15) == FEMCY_PERSIST_DEFAULT_VARIANT) FEMCY_PERSIST(DM_, SPW_, RJ_, FEMCY_PERSIST_DEFAULT_VARIANT); \
        else if ((var & 15) == FEMCY_PERSIST_ALT_VARIANT) FEMCY_PERSIST(DM_, SPW_, RJ_, FEMCY_PERSIST_ALT_VARIANT);   \
        else if ((var & 15) == 0) FEMCY_PERSIST(DM_, SPW_, RJ_, 0);                                                   \
        else { set_error("persistent PCG variant %d is not in this build (%d, %d, 0; 0..7 and 14 with "               \
                         "-DFEMCY_PERSIST_ALL_VARIANTS)", var, FEMCY_PERSIST_DEFAULT_VARIANT, FEMCY_PERSIST_ALT_VARIANT); \
               return FEMCY_EINVAL; }
#endif
        // register-resident block rows per slice: what 512 VGPRs per lane hold next to the vectors and the streaming
        // buffers (dm 3: 5 rows x 3 slices or 3 rows x 4 slices of 19 registers each)
        if (multi) {
            // across ranks: the default variant, 3 x 3 blocks (the agreement checked dm == 3); register rows as in the
            // single-rank kernel of the same shape
            FEMCY_REQUIRE((var & 15) == FEMCY_PERSIST_DEFAULT_VARIANT, "the multi-rank persistent PCG exists for the default variant only");
            if (c->dm == 3) {
                FEMCY_REQUIRE(SPW <= 4, "the multi-rank persistent PCG of 3 x 3 blocks exists for up to four slices per wave");
                if (SPW == 3) { FEMCY_PERSIST_M(3, 3, 4, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
                else { FEMCY_PERSIST_M(3, 4, 3, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
            } else if (SPW == 3) { FEMCY_PERSIST_M(2, 3, 5, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
            else if (SPW == 4) { FEMCY_PERSIST_M(2, 4, 5, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
            else if (SPW == 6) { FEMCY_PERSIST_M(2, 6, 3, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
            else { FEMCY_PERSIST_M(2, 8, 2, FEMCY_PERSIST_DEFAULT_VARIANT, true); }
        } else
#ifdef FEMCY_PERSIST_ONLY_334     // compile-time experiments: one shape only
        if (c->dm == 3 && SPW == 3 && c->opt_persist_rj == 4) { FEMCY_PERSIST_V(3, 3, 4) } else return FEMCY_OK;
#else
        if (c->dm == 3 && SPW == 3) {
            if (c->opt_persist_rj == 5) { FEMCY_PERSIST_V(3, 3, 5) }
            else if (c->opt_persist_rj == 4) { FEMCY_PERSIST_V(3, 3, 4) }
#ifdef FEMCY_PERSIST_PIPE
            else if (c->opt_persist_rj == 2) { FEMCY_PERSIST_V(3, 3, 2) }
#endif
            else { FEMCY_PERSIST_V(3, 3, 0) }
        } else if (c->dm == 3 && SPW == 4) {
            if (c->opt_persist_rj) { FEMCY_PERSIST_V(3, 4, 3) } else { FEMCY_PERSIST_V(3, 4, 0) }
        } else if (c->dm == 3) {
            // round 6: 5 .. 7 slices per wave for 3 x 3 blocks (up to 7 168 slices = 1.37 M DOF: the C3D10 plate at k = 8 keeps
            // one launch per solve).  A lane's five vectors take 30 registers per slice: 150 .. 210 of the 512, so one block
            // row per slice (19 registers) at five / six slices and none at seven is all that stays in registers; default
            // variant only.  Every shape is held to the three-launch loop's iterates by tests/test_gpu_pcg_persist.py, and
            // tools/check_exec_joins.py (a CPU test) guards the compiled code against the miscompile these shapes first
            // ran into (the comment at the initial loads of b and M)
            FEMCY_REQUIRE((var & 15) == FEMCY_PERSIST_DEFAULT_VARIANT, "5 .. 7 slices per wave exist for the default variant of the persistent PCG only");
            // (registers, hipcc 7.0: <3,5,1> / <3,6,1> up to 486 without spills, <3,5,2> and <3,7,1> spill 12 / 32, <3,7,0> 384)
            if (SPW == 5) { if (c->opt_persist_rj) FEMCY_PERSIST(3, 5, 1, FEMCY_PERSIST_DEFAULT_VARIANT); else FEMCY_PERSIST(3, 5, 0, FEMCY_PERSIST_DEFAULT_VARIANT); }
            else if (SPW == 6) { if (c->opt_persist_rj) FEMCY_PERSIST(3, 6, 1, FEMCY_PERSIST_DEFAULT_VARIANT); else FEMCY_PERSIST(3, 6, 0, FEMCY_PERSIST_DEFAULT_VARIANT); }
            else { FEMCY_PERSIST(3, 7, 0, FEMCY_PERSIST_DEFAULT_VARIANT); }
        } else if (SPW == 3) {
            if (c->opt_persist_rj) { FEMCY_PERSIST_V(2, 3, 5) } else { FEMCY_PERSIST_V(2, 3, 0) }
        } else if (SPW == 4) {
            if (c->opt_persist_rj) { FEMCY_PERSIST_V(2, 4, 5) } else { FEMCY_PERSIST_V(2, 4, 0) }
        } else if (SPW == 6) {
            if (c->opt_persist_rj) { FEMCY_PERSIST_V(2, 6, 3) } else { FEMCY_PERSIST_V(2, 6, 0) }
        } else {
            if (c->opt_persist_rj) { FEMCY_PERSIST_V(2, 8, 2) } else { FEMCY_PERSIST_V(2, 8, 0) }
        }
#endif
#undef FEMCY_PERSIST_V
#undef FEMCY_PERSIST
#undef FEMCY_PERSIST_M
        if (launched) timing_end(c, tp);
        FEMCY_HIP(hipGetLastError());
        return FEMCY_OK;
    };
    const int lrc = setup_and_launch();
    if (!multi && (lrc || not_resident)) return lrc;
    if (multi) {
        // every rank learns whether the solve completed EVERYWHERE (a time-out on one rank leaves the others'
        // iterates unusable as well); this collective is also what keeps a fast rank's next solve from writing into
        // mailboxes a slow rank is still reading
        FEMCY_HIP(hipMemcpyAsync(c->h_state, c->d_state, sizeof(PcgState), hipMemcpyDeviceToHost, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        double bad = (lrc != FEMCY_OK || !launched || c->h_state->done == 3) ? 1.0 : 0.0;
        FEMCY_HIP(hipMemcpyAsync(c->d_commbuf, &bad, sizeof(double), hipMemcpyHostToDevice, c->stream));
        int rc = comm_allreduce_sum(c, c->d_commbuf, 1);
        if (rc) return rc;
        FEMCY_HIP(hipMemcpyAsync(&bad, c->d_commbuf, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        if (lrc) return lrc;                                      // (after the vote: nobody is left alone in it)
        if (bad != 0.0) {
            c->persist_multi_failed = true;
            c->persist_multi_fallbacks = 0;
            c->timing.barrier_timeouts++;
            return FEMCY_OK;                                      // *handled stays false: the RCCL loop redoes the solve
        }
    }
    *handled = true;
    return FEMCY_OK;
}

// ---- ceiling probes (femcy_probe_stream / femcy_probe_exchange)
// mode 0: the persistent kernel's launch shape (one workgroup of four waves per CU), 8 loads of 16 bytes in flight per
// lane; mode 1: the same with non-temporal loads; mode 2: 8 workgroups per CU (what the chip streams at full
// occupancy); mode 3: mode 2 + nt; modes 4 / 5: mode 0 with 16 / 32 loads in flight per lane (how much of the gap
// between modes 0 and 2 is memory-level parallelism); modes 6 / 7: modes 4 / 5 non-temporal
int probe_stream(Ctx* c, int64_t bytes, int32_t reps, int32_t mode, double* us_per_pass, int64_t* bytes_per_pass) {
    FEMCY_REQUIRE(bytes >= (1 << 20) && reps >= 1 && mode >= 0 && mode <= 16, "probe_stream: bytes >= 1 MiB, reps >= 1, mode 0..16");
    // modes 14 / 15 / 16: 2 / 3 / 4 workgroups per CU (8 / 12 / 16 waves), 8 loads in flight, default policy
    const int per_cu = (mode == 2 || mode == 3) ? 8 : (mode >= 14 ? mode - 12 : 1);
    const int64_t n16 = bytes / 16;
    // modes 8 / 9 / 10: the LDS-DMA path with 8 / 16 / 32 KiB in flight per wave; 11 / 12 / 13: the same non-temporal
    const bool dma = mode >= 8 && mode <= 13;
    const int nt = (mode == 1 || mode == 3 || mode == 6 || mode == 7 || mode >= 11) ? 1 : 0;
    const int unroll = dma ? (8 << ((mode - 8) % 3)) : ((mode == 4 || mode == 6) ? 16 : ((mode == 5 || mode == 7) ? 32 : 8));
    int G = (c->persist_cus / PNX) * PNX * per_cu;
    FEMCY_REQUIRE(G >= PNX, "device reports %d compute units", c->persist_cus);
    const int64_t tile = (int64_t)PBS * unroll;
    G = (int)std::max<int64_t>(PNX, std::min<int64_t>(G, (n16 / tile) / PNX * PNX));   // >= one tile per workgroup
    const int64_t chunk = (n16 / G) / tile * tile;
    FEMCY_REQUIRE(chunk > 0, "probe_stream: %lld bytes are less than one tile per workgroup", (long long)bytes);
    if (!c->d_probe || c->probe_cap < bytes) {
        if (c->d_probe) (void)hipFree(c->d_probe);
        c->d_probe = nullptr;
        c->probe_cap = bytes;
        FEMCY_HIP(dmalloc(&c->d_probe, (size_t)bytes + 64));
        FEMCY_HIP(hipMemsetAsync(c->d_probe, 0, (size_t)bytes, c->stream));
    }
    // dynamic LDS sets the residency: one workgroup per CU as the solver, or exactly per_cu of them
    const size_t lds = per_cu == 1 ? (size_t)(c->small_max_lds - 1024)
                                   : (per_cu < 8 ? (size_t)(c->small_max_lds / per_cu - 2048) : 0);
    hipEvent_t e0, e1;
    FEMCY_HIP(hipEventCreate(&e0));
    FEMCY_HIP(hipEventCreate(&e1));
#define FEMCY_PROBE(U_, REPS_)                                                                                     \
    do {                                                                                                           \
        const void* fn = reinterpret_cast<const void*>(&k_probe_stream<U_>);                                       \
        if (lds > 48 * 1024) FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_probe_stream<U_>, dim3(G), dim3(PBS), lds, c->stream, (const double2*)c->d_probe, n16, \
                           (int)(REPS_), nt, c->d_part1);                                                          \
    } while (0)
#define FEMCY_PROBE_DMA(U_, REPS_)                                                                                 \
    do {                                                                                                           \
        const void* fn = reinterpret_cast<const void*>(&k_probe_stream_lds<U_>);                                   \
        if (lds > 48 * 1024) FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_probe_stream_lds<U_>, dim3(G), dim3(PBS), lds, c->stream, (const double2*)c->d_probe, n16, \
                           (int)(REPS_), nt, c->d_part1);                                                          \
    } while (0)
    for (int pass = 0; pass < 2; ++pass) {                       // warm-up launch (2 passes), then the timed one
        if (pass) FEMCY_HIP(hipEventRecord(e0, c->stream));
        const int r = pass ? reps : 2;
        if (dma) {
            if (unroll == 8) FEMCY_PROBE_DMA(8, r); else if (unroll == 16) FEMCY_PROBE_DMA(16, r); else FEMCY_PROBE_DMA(32, r);
        } else if (unroll == 8) FEMCY_PROBE(8, r); else if (unroll == 16) FEMCY_PROBE(16, r); else FEMCY_PROBE(32, r);
    }
#undef FEMCY_PROBE_DMA
#undef FEMCY_PROBE
    FEMCY_HIP(hipEventRecord(e1, c->stream));
    FEMCY_HIP(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    FEMCY_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    FEMCY_HIP(hipGetLastError());
    *us_per_pass = (double)ms * 1e3 / reps;
    if (bytes_per_pass) *bytes_per_pass = chunk * G * 16;
    return FEMCY_OK;
}

// one grid-wide exchange of an 8-byte value per workgroup (publish, synchronise, every workgroup sums all G), in the
// solver's launch shape; form 0 = counters + data (grid_barrier), 1 = tagged granules
int probe_exchange(Ctx* c, int32_t rounds, int32_t form, double* us_per_exchange) {
    FEMCY_REQUIRE(rounds >= 1 && rounds <= (1 << 20) && (form == 0 || form == 1), "probe_exchange: rounds 1..2^20, form 0 / 1");
    const int G = (c->persist_cus / PNX) * PNX;
    FEMCY_REQUIRE(G >= PNX, "device reports %d compute units", c->persist_cus);
    PersistPcg a{};
    int rc = persist_buffers(c, G, 2, &a);
    if (rc) return rc;
    const size_t lds = (size_t)(c->small_max_lds - 1024);
    const void* fn = form ? reinterpret_cast<const void*>(&k_probe_exchange<1>) : reinterpret_cast<const void*>(&k_probe_exchange<0>);
    if (lds > 48 * 1024) FEMCY_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FEMCY_REQUIRE(coresident(c, fn, PBS, lds, G), "probe_exchange: %d workgroups cannot be co-resident", G);
    hipEvent_t e0, e1;
    FEMCY_HIP(hipEventCreate(&e0));
    FEMCY_HIP(hipEventCreate(&e1));
    double us[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {                      // rounds and 2 x rounds: the difference is launch-free
        const int r = rounds * (pass + 1);
        if ((rc = persist_buffers(c, G, 2, &a))) return rc;
        FEMCY_HIP(hipEventRecord(e0, c->stream));
        if (form) hipLaunchKernelGGL(k_probe_exchange<1>, dim3(G), dim3(PBS), lds, c->stream, a, r, c->d_part2);
        else hipLaunchKernelGGL(k_probe_exchange<0>, dim3(G), dim3(PBS), lds, c->stream, a, r, c->d_part2);
        FEMCY_HIP(hipEventRecord(e1, c->stream));
        FEMCY_HIP(hipStreamSynchronize(c->stream));
        float ms = 0.f;
        FEMCY_HIP(hipEventElapsedTime(&ms, e0, e1));
        us[pass] = (double)ms * 1e3;
        double res[2];
        FEMCY_HIP(hipMemcpy(res, c->d_part2, sizeof(res), hipMemcpyDeviceToHost));
        const double want = (double)r * (G * (G - 1) / 2.0) + (double)G * (r * (r - 1.0) / 2.0);
        if (res[1] != 0.0 || res[0] != want) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            set_error("probe_exchange: form %d failed (flag %g, sum %.17g, expected %.17g)", form, res[1], res[0], want);
            return FEMCY_EHIP;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    FEMCY_HIP(hipGetLastError());
    *us_per_exchange = (us[1] - us[0]) / rounds;
    return FEMCY_OK;
}

// collective: every rank launches the single-wave probe at the same time (the caller synchronises the ranks first);
// us_per_round = one mailbox all-to-all + poll between the ranks' kernels
int probe_mailbox(Ctx* c, int32_t rounds, double* us_per_round) {
    FEMCY_REQUIRE(c->comm && c->d_mbox && c->d_peer_tab, "femcy_comm_mailbox_import must come first");
    FEMCY_REQUIRE(c->persist_multi_local, "the mailboxes of this communicator are not usable (femcy_comm_mailbox_import)");
    FEMCY_REQUIRE(rounds >= 1 && rounds <= (1 << 19), "probe_mailbox: rounds 1..2^19");
    PersistPcg a{};
    a.mbox = c->d_mbox;
    a.peer = c->d_peer_tab;
    a.rank = c->rank;
    a.nranks = c->nranks;
    a.xspin_limit = (uint32_t)std::min<uint64_t>((uint64_t)c->barrier_spin_limit * 16, 1u << 30);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
    int rc = FEMCY_OK;
    double ticks = 0.0;
    {
        c->solve_serial = (c->solve_serial % 4095) + 1;
        a.tagbase = c->solve_serial << 20;
        const int r = rounds + 1;                                // round 0 is not timed (launch skew between the ranks)
        hipLaunchKernelGGL(k_probe_mailbox, dim3(1), dim3(64), 0, c->stream, a, r, c->d_part2);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = FEMCY_EHIP;
        double res[3] = {0, 0, 0};
        if (!rc && hipMemcpy(res, c->d_part2, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) rc = FEMCY_EHIP;
        const double want = (double)r * (c->nranks * (c->nranks + 1) / 2.0);
        if (!rc && (res[1] != 0.0 || res[0] != want)) {
            set_error("probe_mailbox: a poll timed out or a value was wrong (flag %g, sum %.17g, expected %.17g)", res[1], res[0], want);
            rc = FEMCY_ECOMM;
        }
        ticks = res[2];
        // the ranks leave together: nobody's next solve may write entries a slow rank still polls for
        double flag = rc ? 1.0 : 0.0;
        (void)hipMemcpy(c->d_commbuf, &flag, sizeof(double), hipMemcpyHostToDevice);
        int rc2 = comm_allreduce_sum(c, c->d_commbuf, 1);
        if (!rc) rc = rc2;
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(&flag, c->d_commbuf, sizeof(double), hipMemcpyDeviceToHost);
        if (!rc && flag != 0.0) {
            set_error("probe_mailbox: failed on another rank");
            rc = FEMCY_ECOMM;
        }
    }
    if (rc == FEMCY_EHIP) set_error("probe_mailbox: a HIP call failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc) return rc;
    *us_per_round = ticks / (double)khz * 1e3 / rounds;
    return FEMCY_OK;
}

// bytes of the matrix the persistent kernel streams per iteration on this context (stored - register / LDS rows), 0
// when the system would not take the persistent path
int64_t persist_streamed_bytes(Ctx* c) {
    PersistShape sh;
    if (!persist_shape(c, &sh) || c->nslices < sh.G) return 0;
    const int G = sh.G, DD = c->dm * c->dm, lds_rows = sh.lds_rows, rj = sh.rj;
    const int64_t row_bytes = (int64_t)(DD * 8 + 4) * 64;
    // every slice keeps min(L, rj) rows in registers, and the LDS rows of a wave are full whenever its slices have
    // more rows than rj (true for every mesh that takes this path)
    int64_t reg_rows = 0;
    for (int32_t s = 0; s < c->nslices; ++s) reg_rows += std::min<int64_t>(c->h_slice_len[s], rj);
    const int64_t lds_total = std::min<int64_t>((int64_t)G * 4 * lds_rows, c->stored_rows - reg_rows);
    return (c->stored_rows - reg_rows - lds_total) * row_bytes;
}

}  // namespace femcy
