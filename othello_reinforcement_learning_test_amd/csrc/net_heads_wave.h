// net_heads_wave.h -- policy and value heads computed by ONE WAVE for one position (fp32 VALU, <1 % of the FLOPs),
// shared by the wave-per-position trunk kernels (net_f32.hip, net_h3.hip).  Reference: net.py:62-136.
#pragma once
#include "net.h"

namespace oth {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// `src` = address of channel 0 of THIS LANE's cell (lane = cell index; lanes >= BS*BS pass any valid cell), channel ch
// at src[ch * plane_stride]; scratch: 192 floats of LDS private to the wave; pfc_wt / vfc1_wt: FC weights transposed
// on the host ([2*cells][cells+1], [cells][256]) so that lanes read consecutive outputs.
template <int F, int BS>
__device__ __forceinline__ void heads_wave(const HeadParams& hp, const float* __restrict__ pfc_wt,
                                           const float* __restrict__ vfc1_wt, const float* src, int plane_stride,
                                           float* scratch, int lane, float* __restrict__ lp, float* __restrict__ vout) {
    constexpr int CELLS = BS * BS, NP = CELLS + 1;
    {   // 1x1 convs (+ folded BN) + ReLU: lane = cell
        float a0 = 0.f, a1 = 0.f, av = 0.f;
        for (int ch = 0; ch < F; ++ch) {
            const float x = src[ch * plane_stride];
            a0 = fmaf(x, hp.pconv_w[ch * 2 + 0], a0);
            a1 = fmaf(x, hp.pconv_w[ch * 2 + 1], a1);
            av = fmaf(x, hp.vconv_w[ch], av);
        }
        scratch[lane] = fmaxf(a0 + hp.pconv_b[0], 0.f);        // flatten order (channel, cell): net.py:88
        scratch[64 + lane] = fmaxf(a1 + hp.pconv_b[1], 0.f);
        scratch[128 + lane] = fmaxf(av + hp.vconv_b[0], 0.f);
    }
    // policy FC + log_softmax: lane handles outputs lane and lane + 64
    float l0 = -INFINITY, l1 = -INFINITY;
    {
        float s0 = lane < NP ? hp.pfc_b[lane] : 0.f;
        float s1 = lane + 64 < NP ? hp.pfc_b[lane + 64] : 0.f;
        for (int i = 0; i < 2 * CELLS; ++i) {
            const float x = scratch[(i / CELLS) * 64 + (i % CELLS)];
            if (lane < NP) s0 = fmaf(pfc_wt[(size_t)i * NP + lane], x, s0);
            if (lane + 64 < NP) s1 = fmaf(pfc_wt[(size_t)i * NP + lane + 64], x, s1);
        }
        if (lane < NP) l0 = s0;
        if (lane + 64 < NP) l1 = s1;
    }
    const float m = wave_max(fmaxf(l0, l1));
    const float se = wave_sum((lane < NP ? expf(l0 - m) : 0.f) + (lane + 64 < NP ? expf(l1 - m) : 0.f));
    const float lse = logf(se);
    if (lane < NP) lp[lane] = l0 - m - lse;
    if (lane + 64 < NP) lp[lane + 64] = l1 - m - lse;
    // value FC1 (256 outputs: 4 per lane) + ReLU + FC2 + tanh
    float h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = hp.vfc1_b[lane + 64 * j];
    for (int i = 0; i < CELLS; ++i) {
        const float x = scratch[128 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = fmaf(vfc1_wt[(size_t)i * 256 + lane + 64 * j], x, h[j]);
    }
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) part = fmaf(hp.vfc2_w[lane + 64 * j], fmaxf(h[j], 0.f), part);
    const float tot = wave_sum(part);
    if (lane == 0) *vout = tanhf(tot + hp.vfc2_b[0]);
}

}  // namespace oth
