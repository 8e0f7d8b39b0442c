// net_heads_wave.h -- policy and value heads computed by ONE WAVE for one position (fp32 VALU, <1 % of the FLOPs),
// shared by the wave-per-position trunk kernels (net_f32.hip, net_h3.hip).  Reference: net.py:62-136.
#pragma once
#include "net.h"
#include "wave_bfly.h"

namespace oth {

// xor-butterfly reductions over the wave: register-to-register (wave_bfly.h), bit-identical to the __shfl_xor loops they replaced
__device__ __forceinline__ float wave_sum(float v) { return bfly_sum_f32(v); }
__device__ __forceinline__ float wave_max(float v) { return bfly_max_f32(v); }

// `src[p]` = address of channel 0 of THIS LANE's cell of position p (lane = cell index; lanes >= BS*BS pass any valid
// cell), channel ch at src[p][ch * plane_stride]; scratch: 192 * P floats of LDS private to the wave; pfc_wt / vfc1_wt: FC
// weights transposed on the host ([2*cells][cells+1], [cells][256]) so that lanes read consecutive outputs.
//
// P positions at once: every FC weight is loaded once and used for all of them, and the loops run in batches of U
// independent loads (16, or 12 where the trip count asks for it) followed by their FMAs, one accumulator per position.  The first version took
// one position at a time with rolled loops -- one L2 round trip per iteration, and on 8x8 the 65th policy output was a
// 128-iteration loop on lane 0 alone -- and cost 42-55 k cycles per position: 30 % of the 5x64 kernel on 6x6, 48 % of
// the 2x32 kernel (in-kernel stamps, round 3).
// WIDE (round 4, k_trunk_w6): batches of 36 loads instead of 12-16.  A wave alone on its SIMD hides nothing, so every batch
// is one exposed L2 round trip (~700 cycles): on 6x6 the FCs are 18 batches (21 k cycles per wave for two positions, 10 % of
// k_trunk_w6); with 36 per batch they are 6 (8x8: 32 per batch).  Only for kernels with registers to spare; used by k_trunk_w6
// (tried on the 64- / 128-filter builds of k_trunk_h3: no change).
template <int F, int BS, int P, bool WIDE = false>
__device__ __forceinline__ void heads_wave_n(const HeadParams& hp, const float* __restrict__ pfc_wt,
                                             const float* __restrict__ vfc1_wt, const float* const (&src)[P],
                                             int plane_stride, float* scratch, int lane, float* const (&lp)[P],
                                             float* const (&vout)[P], const bool (&live)[P]) {
    constexpr int CELLS = BS * BS, NP = CELLS + 1, NI = 2 * CELLS;
    constexpr int U1 = 16;                                  // channels per batch (F is a multiple of 16)
    constexpr int U2 = WIDE ? (NI % 32 == 0 ? 32 : 36) : (NI % 16 == 0 ? 16 : 12);   // policy FC inputs per batch (72 = 6 x 12 / 2 x 36, 128 = 8 x 16 / 4 x 32)
    static_assert(F % U1 == 0 && NI % U2 == 0, "batch sizes must divide the trip counts");
    static_assert(NP <= 65, "one policy output per lane, plus at most one more");
    {   // 1x1 convs (+ folded BN) + ReLU: lane = cell
        float a0[P], a1[P], av[P];
#pragma unroll
        for (int p = 0; p < P; ++p) a0[p] = a1[p] = av[p] = 0.f;
#pragma unroll 1   // a batch is the unit: unrolled further, hipcc hoists every load and spills
        for (int c0 = 0; c0 < F; c0 += U1) {
            float x[P][U1];
#pragma unroll
            for (int u = 0; u < U1; ++u)
#pragma unroll
                for (int p = 0; p < P; ++p) x[p][u] = src[p][(c0 + u) * plane_stride];
#pragma unroll
            for (int u = 0; u < U1; ++u) {
                const float w0 = hp.pconv_w[(c0 + u) * 2 + 0], w1 = hp.pconv_w[(c0 + u) * 2 + 1], wv = hp.vconv_w[c0 + u];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    a0[p] = fmaf(x[p][u], w0, a0[p]);
                    a1[p] = fmaf(x[p][u], w1, a1[p]);
                    av[p] = fmaf(x[p][u], wv, av[p]);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            scratch[p * 192 + lane] = fmaxf(a0[p] + hp.pconv_b[0], 0.f);        // flatten order (channel, cell): net.py:88
            scratch[p * 192 + 64 + lane] = fmaxf(a1[p] + hp.pconv_b[1], 0.f);
            scratch[p * 192 + 128 + lane] = fmaxf(av[p] + hp.vconv_b[0], 0.f);
        }
    }
    // policy FC: lane = output (lanes >= NP idle); the 65th output of an 8x8 board is a wave reduction over the inputs
    const int ol = lane < NP ? lane : 0;
    float s0[P], sx[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s0[p] = hp.pfc_b[ol];
        sx[p] = 0.f;
    }
#pragma unroll 1   // a batch is the unit: unrolled further, hipcc hoists every load and spills
    for (int i0 = 0; i0 < NI; i0 += U2) {
        float w[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) w[u] = pfc_wt[(size_t)(i0 + u) * NP + ol];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int i = i0 + u;   // rolled outer loop: the feature index is formed at run time
            const int fi = (i / CELLS) * 64 + (i % CELLS);
#pragma unroll
            for (int p = 0; p < P; ++p) s0[p] = fmaf(w[u], scratch[p * 192 + fi], s0[p]);
        }
    }
    if constexpr (NP > 64) {
#pragma unroll
        for (int k = 0; k < (NI + 63) / 64; ++k) {
            const int i = lane + 64 * k;
            if (i < NI) {
                const float w = pfc_wt[(size_t)i * NP + 64];
                const int fi = (i / CELLS) * 64 + (i % CELLS);
#pragma unroll
                for (int p = 0; p < P; ++p) sx[p] = fmaf(w, scratch[p * 192 + fi], sx[p]);
            }
        }
    }
    // value FC1 (256 outputs: 4 per lane)
    float h[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) h[p][0] = h[p][1] = h[p][2] = h[p][3] = 0.f;
    // One output column per pass (j), batches of U2 rows -- the shape of the policy FC above.
    //
    // Round 3 shipped this form because a batch of U3 rows x 4 columns gave wrong value outputs at ~1-2 % of the positions,
    // different ones on every launch, in k_trunk_h3<32, 8, 1, 4> with two workgroups per CU.  Round 4 found the cause: for
    // the 4-column batch hipcc's SLP vectoriser emits `v_pk_fma_f32 ... op_sel:[0,1,0]` -- the HIGH dword of the x pair
    // routed to the low lane -- and on gfx950 that operand form returns wrong results whenever ANOTHER wave of the same SIMD
    // is issuing MFMAs at the time (tools/probes/probe_pk_opsel.hip: 24-29 % of wave-results wrong beside an MFMA partner,
    // none beside an idle / LDS / VALU / VMEM partner, none ever for the low-dword broadcast `op_sel_hi:[1,0,1]` or for
    // plain pairs; in this kernel, eight builds differing in nothing but that operand form: profiles/r04_heads_batch4_variants.txt --
    // the batched form itself was removed from this file in round 5, it is in the history at 274f240).  The 64-filter builds had the same instructions and never failed: they
    // run one wave per SIMD.  The library is therefore built with -fno-slp-vectorize (hipcc emits no packed fp32 of its
    // own; measured equal or faster), net_epilogue.h's packed arithmetic uses plain pairs, and
    // tools/check_mfma_hazards.py fails the build on any packed-fp32 instruction with a high-to-low operand select.
    constexpr int U4 = WIDE ? (CELLS % 32 == 0 ? 32 : 36) : (CELLS % 16 == 0 ? 16 : 12);
    static_assert(CELLS % U4 == 0, "batch size must divide the trip count");
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        float hj[P];
#pragma unroll
        for (int p = 0; p < P; ++p) hj[p] = hp.vfc1_b[lane + 64 * j];
#pragma unroll 1
        for (int i0 = 0; i0 < CELLS; i0 += U4) {
            float w[U4];
#pragma unroll
            for (int u = 0; u < U4; ++u) w[u] = vfc1_wt[(size_t)(i0 + u) * 256 + lane + 64 * j];
#pragma unroll
            for (int u = 0; u < U4; ++u)
#pragma unroll
                for (int p = 0; p < P; ++p) hj[p] = fmaf(w[u], scratch[p * 192 + 128 + i0 + u], hj[p]);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {   // h[p][j] with a run-time j: select, do not index (an indexed array goes to scratch memory)
            h[p][0] = j == 0 ? hj[p] : h[p][0];
            h[p][1] = j == 1 ? hj[p] : h[p][1];
            h[p][2] = j == 2 ? hj[p] : h[p][2];
            h[p][3] = j == 3 ? hj[p] : h[p][3];
        }
    }
    float w2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w2[j] = hp.vfc2_w[lane + 64 * j];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        // log_softmax over the NP logits, FC2 + tanh
        const float l0 = lane < NP ? s0[p] : -INFINITY;
        float l64 = -INFINITY;
        if constexpr (NP > 64) l64 = wave_sum(sx[p]) + hp.pfc_b[64];
        const float m = fmaxf(wave_max(l0), l64);
        float e = lane < NP ? expf(l0 - m) : 0.f;
        if (NP > 64 && lane == 0) e += expf(l64 - m);
        const float lse = logf(wave_sum(e));
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) part = fmaf(w2[j], fmaxf(h[p][j], 0.f), part);
        const float tot = wave_sum(part);
        if (live[p]) {
            if (lane < NP && lane < 64) lp[p][lane] = l0 - m - lse;
            if (NP > 64 && lane == 0) lp[p][64] = l64 - m - lse;
            if (lane == 0) *vout[p] = tanhf(tot + hp.vfc2_b[0]);
        }
    }
}

// one position (net_f32.hip)
template <int F, int BS>
__device__ __forceinline__ void heads_wave(const HeadParams& hp, const float* __restrict__ pfc_wt,
                                           const float* __restrict__ vfc1_wt, const float* src, int plane_stride,
                                           float* scratch, int lane, float* __restrict__ lp, float* __restrict__ vout) {
    const float* const srcs[1] = {src};
    float* const lps[1] = {lp};
    float* const vs[1] = {vout};
    const bool live[1] = {true};
    heads_wave_n<F, BS, 1>(hp, pfc_wt, vfc1_wt, srcs, plane_stride, scratch, lane, lps, vs, live);
}

}  // namespace oth
