// net_heads.h -- policy and value heads (fp32 VALU), shared by the generic and the MFMA trunk kernels.
#pragma once
#include "net.h"
#include "wave_bfly.h"

namespace oth {

// ------------------------------------------------------------------------------------------------
// heads (fp32 VALU), shared by both trunk kernels.  `act` = final trunk activation of ONE position
// in LDS as [64 cells][F] floats with row stride `ld` floats; scratch >= 128+64+256+72 floats.
// All threads of the block (256 or 512) must call this.  (net.py:83-96 policy, net.py:119-136 value)
// ------------------------------------------------------------------------------------------------
__device__ inline void heads_forward(const HeadParams& hp, int F, const float* act, int ld, float* scratch,
                              float* logp65, float* v1) {
    const int t = threadIdx.x;
    float* pf = scratch;         // [2][64] policy features, flatten order (channel, cell)
    float* vf = scratch + 128;   // [64]
    float* h1 = scratch + 192;   // [256]
    float* lg = scratch + 448;   // [65] logits (+ 2 reduction slots)
    if (t < 192) {
        const int cell = t & 63, ch = t >> 6;  // ch 0,1: policy planes; 2: value plane
        const float* a = act + cell * ld;
        float acc = 0.f;
        if (ch < 2) {
            for (int i = 0; i < F; ++i) {
                const int ii = (i + cell) & (F - 1);  // rotate the start per lane: conflict-free LDS rows
                acc = fmaf(a[ii], hp.pconv_w[ii * 2 + ch], acc);
            }
            acc += hp.pconv_b[ch];
            pf[ch * 64 + cell] = acc > 0.f ? acc : 0.f;
        } else {
            for (int i = 0; i < F; ++i) {
                const int ii = (i + cell) & (F - 1);
                acc = fmaf(a[ii], hp.vconv_w[ii], acc);
            }
            acc += hp.vconv_b[0];
            vf[cell] = acc > 0.f ? acc : 0.f;
        }
    }
    __syncthreads();
    if (t < 256) {   // value fc1: 256 outputs, one per thread (a 512-thread caller's upper half only takes the barriers)
        const float* w = hp.vfc1_w + t * 64;
        float acc = hp.vfc1_b[t];
        for (int i = 0; i < 64; ++i) acc = fmaf(w[i], vf[i], acc);
        h1[t] = acc > 0.f ? acc : 0.f;
    }
    if (t < 65) {  // policy fc
        const float* w = hp.pfc_w + t * 128;
        float acc = hp.pfc_b[t];
        for (int i = 0; i < 128; ++i) acc = fmaf(w[i], pf[i], acc);
        lg[t] = acc;
    }
    __syncthreads();
    if (t < 64) {  // wave 0: log_softmax over 65 logits and the fc2 dot product
        float m = fmaxf(lg[t], t == 0 ? lg[64] : -INFINITY);
        m = bfly_max_f32(m);
        float s = expf(lg[t] - m) + (t == 0 ? expf(lg[64] - m) : 0.f);
        s = bfly_sum_f32(s);
        const float lse = logf(s);
        logp65[t] = lg[t] - m - lse;
        if (t == 0) logp65[64] = lg[64] - m - lse;
        float acc = 0.f;
        for (int i = t; i < 256; i += 64) acc = fmaf(hp.vfc2_w[i], h1[i], acc);
        acc = bfly_sum_f32(acc);
        if (t == 0) *v1 = tanhf(acc + hp.vfc2_b[0]);
    }
    __syncthreads();
}


}  // namespace oth
