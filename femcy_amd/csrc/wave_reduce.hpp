// Wavefront reductions on the DPP network of gfx950 (quad swaps, row mirrors, row broadcasts: six steps of ~8 clocks)
// instead of twelve dependent ds_bpermute round trips through the LDS crossbar (what __shfl_down of a double compiles
// to).  The order of the additions is fixed; the result (lane 63's) is returned in EVERY lane.  All 64 lanes must be
// active.
#pragma once
#include <hip/hip_runtime.h>

namespace femcy {

template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_move(double v) {   // lanes of rows outside ROWS receive 0.0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xf>(v);     // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);     // row_mirror: every lane of a row of 16 holds the row's sum
    v += dpp_move<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add the 0.0 of `old`)
    v += dpp_move<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
    return lane63(v);
}
// v >= 0 in every lane (absolute values, +inf standing for NaN): masked rows contribute the 0.0 of `old`
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_move<0xB1, 0xf>(v));
    v = fmax(v, dpp_move<0x4E, 0xf>(v));
    v = fmax(v, dpp_move<0x141, 0xf>(v));
    v = fmax(v, dpp_move<0x140, 0xf>(v));
    v = fmax(v, dpp_move<0x142, 0xa>(v));
    v = fmax(v, dpp_move<0x143, 0xc>(v));
    return lane63(v);
}

}  // namespace femcy
