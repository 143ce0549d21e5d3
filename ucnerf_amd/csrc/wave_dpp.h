// Wave64 reductions and prefix sums on the DPP path of the VALU (row_shr 1 / 2 / 4 / 8 inside a row of 16 lanes, then
// row_bcast15 / row_bcast31 across the rows): six dependent 4-cycle VALU steps.  The `__shfl_*` forms of the same loops
// compile to six dependent ds_bpermute_b32 round trips through the LDS crossbar (k_composite issued 68 of them per ray).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or0(int v) {          // the DPP-selected lane's value, 0 where there is none
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or0(float v) { return __builtin_bit_cast(float, dpp_or0<CTRL, ROW_MASK>(__builtin_bit_cast(int, v))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t v) { return (uint32_t)dpp_or0<CTRL, ROW_MASK>((int)v); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or0(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)dpp_or0<CTRL, ROW_MASK>((int)(uint32_t)b), hi = (uint32_t)dpp_or0<CTRL, ROW_MASK>((int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// inclusive prefix sum across the 64 lanes (T = float, double, uint32_t)
template <class T>
__device__ __forceinline__ T wave_scan_dpp(T v) {
    v += dpp_or0<0x111, 0xf>(v);        // row_shr:1
    v += dpp_or0<0x112, 0xf>(v);        // row_shr:2
    v += dpp_or0<0x114, 0xf>(v);        // row_shr:4
    v += dpp_or0<0x118, 0xf>(v);        // row_shr:8   -> inclusive within each row of 16
    v += dpp_or0<0x142, 0xa>(v);        // row_bcast15 into rows 1 and 3
    v += dpp_or0<0x143, 0xc>(v);        // row_bcast31 into rows 2 and 3
    return v;
}
// the value of the lane below (0 in lane 0): inclusive -> exclusive
template <class T>
__device__ __forceinline__ T wave_shift_up1(T v) { return dpp_or0<0x138, 0xf>(v); }   // wave_shr:1
template <class T>
__device__ __forceinline__ T wave_last(T v);                                            // lane 63's value in every lane
template <>
__device__ __forceinline__ float wave_last<float>(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
template <>
__device__ __forceinline__ uint32_t wave_last<uint32_t>(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
template <>
__device__ __forceinline__ double wave_last<double>(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T wave_sum_dpp(T v) { return wave_last<T>(wave_scan_dpp<T>(v)); }

// maximum over the wave (non-negative or any floats: the identity is the lane's own value)
__device__ __forceinline__ float wave_max_dpp(float v) {
#define UCN_MAX_STEP(CTRL, RM) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RM, 0xf, false)))
    UCN_MAX_STEP(0x111, 0xf); UCN_MAX_STEP(0x112, 0xf); UCN_MAX_STEP(0x114, 0xf); UCN_MAX_STEP(0x118, 0xf);
    UCN_MAX_STEP(0x142, 0xa); UCN_MAX_STEP(0x143, 0xc);
#undef UCN_MAX_STEP
    return wave_last<float>(v);
}

// LDS written by some lanes and read by OTHER LANES OF THE SAME WAVE (a wave working in its own LDS region): the writes
// are drained (lgkmcnt) and the compiler may not move LDS accesses across, but no s_barrier -- the other waves of the
// workgroup work on other rays and have nothing to wait for.
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// v + (the value of lane ^ 32): the two halves of a sample's accumulator row meet through one v_permlane32_swap (VALU)
// instead of a ds_bpermute_b32 round trip; the same two addends, so bit-identical to v + __shfl_xor(v, 32).
__device__ __forceinline__ float xor32_sum(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);   // r[0] = {lo, lo}, r[1] = {hi, hi}
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}

}  // namespace
