// net_epilogue.h -- instruction-level helpers of the fused trunks' epilogues (net_wino.hip, net_h3.hip): the fp16 hi/lo
// re-split of an fp32 activation in three instructions per pair of values, packed fp32 arithmetic on register pairs, and
// the in-place skip connection.  Plain VALU inline asm: no MFMA hazards involved (tools/check_mfma_hazards.py looks at
// VALU writes in front of MFMAs only, and none of these results feeds an MFMA directly -- they go through LDS).
#pragma once
#include <stdint.h>

namespace oth {

using f32x4_e = float __attribute__((ext_vector_type(4)));

// two fp32 values -> packed f16 (round to nearest even)
__device__ __forceinline__ uint32_t wpack(float a, float b) {
    using half2v = _Float16 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, half2v{(_Float16)a, (_Float16)b});
}
// packed f16 of (a - hi.lo, b - hi.hi): the low parts of the operand split.  The fp32 difference is exact (hi is a
// rounded to 11 bits), so the only rounding is the final one to f16 -- the same value as converting hi back, subtracting
// and converting again, in two instructions.
__device__ __forceinline__ uint32_t wresid(uint32_t hi, float a, float b) {
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
    return lo;
}

using f32x2 = float __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 whalf(const f32x4_e& v, int h) { return h == 0 ? f32x2{v[0], v[1]} : f32x2{v[2], v[3]}; }
__device__ __forceinline__ void wsethalf(f32x4_e& v, int h, f32x2 x) {
    v[2 * h] = x.x;
    v[2 * h + 1] = x.y;
}
// packed fp32 (two independent IEEE operations per instruction: the results are those of the scalar forms)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {   // a - b
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_nc(f32x2 a, f32x2 b, f32x2 c) {   // a * b - c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_na(f32x2 a, f32x2 b, f32x2 c) {   // c - a * b
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ void pk_add_relu_inplace(f32x2& r, f32x2 t, float clamp) {   // r = min(max(r + t, 0), clamp) in r's registers
    asm("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(t));
    float x = r.x, y = r.y;
    asm("v_med3_f32 %0, %0, 0, %1" : "+v"(x) : "v"(clamp));
    asm("v_med3_f32 %0, %0, 0, %1" : "+v"(y) : "v"(clamp));
    r = f32x2{x, y};
}

}  // namespace oth
