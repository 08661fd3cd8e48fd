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

// min over the wave of a 64-bit unsigned key (arg-min of (distance, index) pairs)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v)
{
    unsigned lo = (unsigned)dpp_mov<CTRL, ROW_MASK>((int)(unsigned)v, (int)(unsigned)v);
    unsigned hi = (unsigned)dpp_mov<CTRL, ROW_MASK>((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32));
    unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o < v ? o : v;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    v = dpp_min_u64<0x111, 0xF>(v);
    v = dpp_min_u64<0x112, 0xF>(v);
    v = dpp_min_u64<0x114, 0xF>(v);
    v = dpp_min_u64<0x118, 0xF>(v);
    v = dpp_min_u64<0x142, 0xA>(v);
    v = dpp_min_u64<0x143, 0xC>(v);
    unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

}  // namespace gpscal
