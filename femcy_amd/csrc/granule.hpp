// Tagged-granule exchange between the workgroups of one persistent launch (k_pcg_persist, k_pcg_small).
// A granule is one naturally aligned 16-byte {f64 value, u64 tag} written by ONE sc1 store (observed untorn on
// gfx950, MI355X_MICROARCH.md "Valid forms": R2); the tag is a launch-wide round number (1, 2, 3, ...; the arrays are
// zeroed before the launch), so a granule validates itself: no counter, no flag, no re-arming.  A workgroup may write
// round k + 1 of an array only after every workgroup has read round k of it -- the callers rotate arrays so that an
// exchange in between guarantees that.  Every sweep is bounded; TAG_POISON releases everybody with the same verdict.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_reduce.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the sc1-only hand-off protocols of the one-launch PCG kernels are validated for gfx942 / gfx950 only"
#endif

namespace femcy {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;     // gfx940+ buffer cache-policy bit 4: sc1 = write-through store / L2-served load
constexpr unsigned long long TAG_POISON = ~0ull;

__device__ __forceinline__ void granule_store(const __amdgpu_buffer_rsrc_t rs, int idx, double v, unsigned long long tag) {
    u32x4 w;
    w.x = (unsigned)__double2loint(v);
    w.y = (unsigned)__double2hiint(v);
    w.z = (unsigned)tag;
    w.w = (unsigned)(tag >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(w, rs, idx * 16, 0, AUX_SC1);
}
// ONE wave sweeps: NV granules per workgroup, stored [G][NV]; lane sweeps workgroups lane, lane + 64, ...; sums /
// maxima in a fixed order (the same in every workgroup).  op[v]: 0 = sum, 1 = max.  Returns false on poison or
// time-out (the caller's workgroup then poisons its own granules, which every other sweep sees).
template <int NV>
__device__ __forceinline__ bool granule_sweep(const __amdgpu_buffer_rsrc_t rs, int base, int G, unsigned long long tag,
                                              uint32_t spin_limit, double (&out)[NV], const int (&op)[NV]) {
    const int lane = threadIdx.x & 63;
    uint32_t spins = 0;
#ifdef FEMCY_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(FEMCY_SWEEP_PRIO);   /* experiment: the sweeping wave ahead of the CU's streaming waves */
#endif
    for (;;) {
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
        bool ok = true, poison = false;
        for (int k = lane; k < G; k += 64) {
            u32x4 w[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) w[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + k * NV + v) * 16, 0, AUX_SC1);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const unsigned long long t = ((unsigned long long)w[v].w << 32) | w[v].z;
                ok = ok && (t == tag);
                poison = poison || (t == TAG_POISON);
                const double val = __hiloint2double((int)w[v].y, (int)w[v].x);
                acc[v] = op[v] ? fmax(acc[v], val) : acc[v] + val;
            }
        }
#ifdef FEMCY_SWEEP_PRIO
        if (__any(poison) || __all(ok)) __builtin_amdgcn_s_setprio(0);
#endif
        if (__any(poison)) return false;
        if (__all(ok)) {
#pragma unroll
            for (int v = 0; v < NV; ++v) out[v] = op[v] ? wave_max(acc[v]) : wave_sum(acc[v]);
            return true;
        }
#ifndef FEMCY_GRANULE_SLEEP
#define FEMCY_GRANULE_SLEEP 1      /* s_sleep units (64 cycles) between sweeps; 0 = poll back to back (measured: no gain) */
#endif
        if (FEMCY_GRANULE_SLEEP) __builtin_amdgcn_s_sleep(FEMCY_GRANULE_SLEEP);
        if (++spins > spin_limit) return false;
    }
}

}  // namespace femcy
