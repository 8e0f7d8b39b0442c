// wave_bfly.h -- the xor-butterfly wave reductions (off = 32, 16, 8, 4, 2, 1) of the heads and the tree kernel WITHOUT the LDS
// crossbar: HIP's __shfl_xor is a ds_bpermute_b32 behind four address instructions, and a reduction is six of them in a
// dependent chain (three such reductions per position were 3.9 k of the 15.7 k cycles of k_trunk_w6's heads, round 6).
// Here every step is register-to-register: v_permlane32_swap / v_permlane16_swap (gfx950) for the two cross-row steps, DPP
// row_ror:8, row_shl:4 + row_shr:4 and two quad_perms inside a row.  SAME pairing as the butterfly, so sums are BIT-IDENTICAL
// to the __shfl_xor form: a step returns two values (x, y) with {x, y} = {v[lane], v[lane ^ OFF]} as a set -- the swaps hand
// both lanes of a pair the same ordered pair -- and every operator used with it is commutative (IEEE add, max, integer add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oth {

template <int OFF>
__device__ __forceinline__ void bfly_pair(uint32_t v, uint32_t& x, uint32_t& y) {
    static_assert(OFF == 32 || OFF == 16 || OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "one butterfly step");
    if constexpr (OFF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        x = r[0];
        y = r[1];
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        x = r[0];
        y = r[1];
    } else if constexpr (OFF == 8) {
        x = v;
        y = __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xf, 0xf, false);       // row_ror:8
    } else if constexpr (OFF == 4) {
        x = v;
        uint32_t t = __builtin_amdgcn_update_dpp(0u, v, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0, 2 (lane + 4)
        y = __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);             // row_shr:4 into banks 1, 3 (lane - 4)
    } else if constexpr (OFF == 2) {
        x = v;
        y = __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);        // quad_perm [2, 3, 0, 1]
    } else {
        x = v;
        y = __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1, 0, 3, 2]
    }
}

#define OTH_BFLY_STEPS(STEP) STEP(32) STEP(16) STEP(8) STEP(4) STEP(2) STEP(1)

__device__ __forceinline__ float bfly_sum_f32(float v) {
#define OTH_STEP(O) { uint32_t x, y; bfly_pair<O>(__float_as_uint(v), x, y); v = __uint_as_float(x) + __uint_as_float(y); }
    OTH_BFLY_STEPS(OTH_STEP)
#undef OTH_STEP
    return v;
}
__device__ __forceinline__ float bfly_max_f32(float v) {
#define OTH_STEP(O) { uint32_t x, y; bfly_pair<O>(__float_as_uint(v), x, y); v = fmaxf(__uint_as_float(x), __uint_as_float(y)); }
    OTH_BFLY_STEPS(OTH_STEP)
#undef OTH_STEP
    return v;
}
__device__ __forceinline__ int bfly_sum_i32(int v) {
#define OTH_STEP(O) { uint32_t x, y; bfly_pair<O>((uint32_t)v, x, y); v = (int)x + (int)y; }
    OTH_BFLY_STEPS(OTH_STEP)
#undef OTH_STEP
    return v;
}
__device__ __forceinline__ int bfly_max_i32(int v) {
#define OTH_STEP(O) { uint32_t x, y; bfly_pair<O>((uint32_t)v, x, y); v = max((int)x, (int)y); }
    OTH_BFLY_STEPS(OTH_STEP)
#undef OTH_STEP
    return v;
}
__device__ __forceinline__ double bfly_max_f64(double v) {
#define OTH_STEP(O) { const uint64_t b = (uint64_t)__double_as_longlong(v); uint32_t xl, yl, xh, yh;                    \
                      bfly_pair<O>((uint32_t)b, xl, yl); bfly_pair<O>((uint32_t)(b >> 32), xh, yh);                    \
                      v = fmax(__longlong_as_double((long long)(((uint64_t)xh << 32) | xl)),                           \
                               __longlong_as_double((long long)(((uint64_t)yh << 32) | yl))); }
    OTH_BFLY_STEPS(OTH_STEP)
#undef OTH_STEP
    return v;
}
#undef OTH_BFLY_STEPS

}  // namespace oth
