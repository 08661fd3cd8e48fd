// wave_reduce.hpp -- 64-lane reductions on the VALU with DPP row shifts
// (row_shr 1,2,4,8 then row_bcast 15 / 31; result in lane 63, broadcast with
// v_readlane).  ds_bpermute-based __shfl reductions go through the LDS pipe and
// made the fused ICP kernel LDS-issue-bound (204 bpermutes per wave); these do not
// touch LDS.  Summation order is fixed, so results are run-to-run reproducible.
#pragma once
#include <hip/hip_runtime.h>

namespace gpscal {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    int lo = dpp_mov<CTRL, ROW_MASK>(0, __double2loint(v));
    int hi = dpp_mov<CTRL, ROW_MASK>(0, __double2hiint(v));
    return v + __hiloint2double(hi, lo);
}

// Sum over the 64 lanes; the value is returned in every lane (wave-uniform).
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add_f64<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add_f64<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add_f64<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add_f64<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of each row = row total
    v = dpp_add_f64<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add_f64<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> lane 63 = total
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_min(int v)
{
    v = min(v, dpp_mov<0x111, 0xF>(v, v));
    v = min(v, dpp_mov<0x112, 0xF>(v, v));
    v = min(v, dpp_mov<0x114, 0xF>(v, v));
    v = min(v, dpp_mov<0x118, 0xF>(v, v));
    v = min(v, dpp_mov<0x142, 0xA>(v, v));
    v = min(v, dpp_mov<0x143, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

}  // namespace gpscal
