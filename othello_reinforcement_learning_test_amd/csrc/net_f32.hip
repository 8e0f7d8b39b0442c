// net_f32.hip -- OthelloResNet forward in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32), for every
// filter count (16 / 32 / 64 / 128) and both board sizes (8x8, 6x6) the reference's configurations use.
//
// Reference: /root/reference/src/model/net.py:139-205 (eval mode; BatchNorm folded at load time); the narrow and
// 6x6 configurations are configs/test.yaml:8-9 (2x16), configs/debug_6x6.yaml (5x64, board 6), tests/test_model.py.
//
// Design (MI355X-first; replaces the VALU kernel that ran at 4.3 TFLOP/s):
//   * ONE WAVE carries P whole positions through the stem, all residual blocks and both heads.  Waves share
//     nothing, so the trunk has no workgroup barrier at all; a workgroup is just WPB waves packed onto a CU.
//   * activations live in LDS as fp32 planes [channel][position][(BS+2) x (BS+2) cells], zero border included:
//     a 3x3 tap is a constant address offset (dy*(BS+2)+dx), never a bounds check -- every ds_read_b32 of the
//     conv loop is `base + immediate`.
//   * GEMM view per layer: D[out channel][cell] += W[out][in, tap] * X[in, tap][cell], on 16x16x4 tiles:
//     A = weights (lane l: row l&15, k = l>>4), B = activations (lane l: k = l>>4, column l&15 = cell of the tile),
//     D: lane l holds column l&15 and rows 4*(l>>4)+r.  For a (4 input channels, tap) step a wave reads T
//     activation dwords and feeds NB*T MFMAs (NB = F/16 row blocks x T tiles of 16 cells), all into independent
//     accumulators (the 16x16x4 form needs >= 2 independent chains to reach its issue rate).
//   * weights stream from L2 in MFMA fragment order, host-packed so that a lane's 9*NB values of a k-step are
//     consecutive 16-byte chunks, coalesced across the wave; the next k-step's chunks are requested before the
//     current one's MFMAs (register double buffer).
//   * accumulators and the fp32 residual stay in registers; the epilogue (bias, skip, ReLU) rewrites the planes
//     in place.  The arithmetic is a k-ordered fp32 fma chain: same rounding class as the reference's fp32 conv.
//   * heads run per wave from the planes (fp32 VALU, <1 % of the FLOPs) with FC weights transposed on the host so
//     that lanes read consecutive outputs.
#include <math.h>
#include <string.h>

#include <vector>

#include "net.h"
#include "net_heads_wave.h"

namespace oth {

using f32x4 = float __attribute__((ext_vector_type(4)));

struct F32Weights {
    float4* d_w = nullptr;       // [layer][k-step][chunk][64 lanes] x 16 B
    float* d_bias = nullptr;     // [layers][F]
    float* d_pfc_wt = nullptr;   // [2*cells][cells+1] (transposed policy FC)
    float* d_vfc1_wt = nullptr;  // [cells][256]      (transposed value FC1)
    std::vector<uint32_t> layer_off;  // in float4 units
};

struct F32Args {
    const float4* w;
    uint32_t layer_off[kMaxTrunkLayers];
    const float* bias;
    int n_layers;  // 1 + 2*blocks
    HeadParams heads;
    const float* pfc_wt;
    const float* vfc1_wt;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <int F, int BS, int P>
struct TrunkGeom {
    static constexpr int NB = F / 16;               // 16-channel row blocks
    static constexpr int CELLS = BS * BS;
    static constexpr int NP = CELLS + 1;            // policy outputs (pass included)
    static constexpr int NC = P * CELLS;            // output cells of a wave
    static constexpr int T = (NC + 15) / 16;        // 16-cell tiles
    static constexpr int PW = BS + 2;               // padded row
    static constexpr int POSW = PW * PW;            // padded plane of one position (words)
    static constexpr int PS = P * POSW;             // plane stride (words)
    static constexpr int NPL = F < 4 ? 4 : F;       // planes (the stem reads 4: self, opp, legal, zero)
    static constexpr int NW = 9 * NB;               // weight values per lane per k-step
    static constexpr int NCH = (NW + 3) / 4;        // 16-byte chunks per lane per k-step
    static constexpr int SCRATCH = 192 * P;         // head scratch (words) per position: pf0[64] pf1[64] vf[64]
    static constexpr int WAVE_WORDS = NPL * PS + SCRATCH;
};

template <int F, int BS, int P, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_trunk_f32(F32Args a, const uint64_t* __restrict__ sb,
                                                        const uint64_t* __restrict__ ob,
                                                        const uint64_t* __restrict__ lgl, int64_t n,
                                                        const int32_t* __restrict__ n_valid,
                                                        float* __restrict__ logp, float* __restrict__ vout) {
    using G = TrunkGeom<F, BS, P>;
    constexpr int NB = G::NB, CELLS = G::CELLS, NP = G::NP, NC = G::NC, T = G::T, PW = G::PW, POSW = G::POSW,
                  PS = G::PS, NCH = G::NCH;
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pos0 = ((int64_t)blockIdx.x * WPB + wave) * P;
    if (pos0 >= nv) return;   // waves are independent: no barrier below
    const int n16 = lane & 15, kg = lane >> 4;
    float* act = lds_f32 + (size_t)wave * G::WAVE_WORDS;
    float* scratch = act + G::NPL * PS;

    // ---- planes: zero everything (borders stay zero for the whole network), then the three input planes
    for (int i = lane; i < G::NPL * PS; i += 64) act[i] = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (pos0 + p < nv && lane < CELLS) {
            const uint64_t b0 = sb[pos0 + p], b1 = ob[pos0 + p], b2 = lgl[pos0 + p];
            const int off = p * POSW + (lane / BS + 1) * PW + (lane % BS + 1);
            act[0 * PS + off] = (b0 >> lane) & 1ULL ? 1.f : 0.f;   // get_tensor_input planes (bitboard.pyx:309-323)
            act[1 * PS + off] = (b1 >> lane) & 1ULL ? 1.f : 0.f;
            act[2 * PS + off] = (b2 >> lane) & 1ULL ? 1.f : 0.f;
        }
    }

    // ---- per-lane tile geometry: padded-plane offset of this lane's cell in tile t (+ its k-group plane for reads)
    int rd_off[T], wr_off[T];
    bool valid[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ci = t * 16 + n16;
        valid[t] = ci < NC;
        const int c = valid[t] ? ci : 0;
        const int p = c / CELLS, r = c % CELLS;
        const int cell = p * POSW + (r / BS + 1) * PW + (r % BS + 1);
        rd_off[t] = cell + kg * PS;          // B operand: input channel 4*kc + kg
        wr_off[t] = cell + 4 * kg * PS;      // D rows: output channel 16*blk + 4*kg + r
    }

    f32x4 acc[NB][T], res[NB][T];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            res[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    for (int layer = 0; layer < a.n_layers; ++layer) {
        const int KC = layer == 0 ? 1 : F / 4;   // k-steps of 4 input channels (stem: 3 planes + a zero plane)
        const float4* wl = a.w + a.layer_off[layer] + lane;   // + (kc*NCH + chunk)*64
        float4 wq[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) wq[c] = wl[(size_t)c * 64];
        float4 bias_q[NB];   // the epilogue's biases, requested before the convolution (one wave per SIMD: a load inside
                             // the epilogue is a bare L2 round trip per row block)
        constexpr bool PREB = NB <= 4;   // at 128 filters the 32 extra live registers cost more than the wait (-4.5 %)
        if constexpr (PREB) {
#pragma unroll
            for (int b = 0; b < NB; ++b) bias_q[b] = *(const float4*)(a.bias + (size_t)layer * F + b * 16 + 4 * kg);
        }
        for (int kc = 0; kc < KC; ++kc) {
            float w[NCH * 4];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                w[4 * c + 0] = wq[c].x; w[4 * c + 1] = wq[c].y; w[4 * c + 2] = wq[c].z; w[4 * c + 3] = wq[c].w;
            }
            {   // request the next k-step's fragments (the last step re-reads its own: harmless)
                const int kn = kc + 1 < KC ? kc + 1 : kc;
#pragma unroll
                for (int c = 0; c < NCH; ++c) wq[c] = wl[((size_t)kn * NCH + c) * 64];
            }
            const float* src = act + kc * 4 * PS;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = (tap / 3 - 1) * PW + (tap % 3 - 1);
                float bv[T];
#pragma unroll
                for (int t = 0; t < T; ++t) bv[t] = src[rd_off[t] + toff];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[b][t] = mfma4(w[tap * NB + b], bv[t], acc[b][t]);
            }
        }
        // ---- epilogue: bias, skip connection (net.py:58-59), ReLU; rewrite the planes in place
        const bool add_res = layer > 0 && (layer & 1) == 0;   // second conv of a block
        const bool set_res = layer == 0 || add_res;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 bias = PREB ? bias_q[b] : *(const float4*)(a.bias + (size_t)layer * F + b * 16 + 4 * kg);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = acc[b][t];
                v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
                if (add_res) v += res[b][t];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                if (set_res) res[b][t] = v;
                acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (valid[t]) {
                    float* dst = act + (b * 16) * PS + wr_off[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * PS] = v[r];
                }
            }
        }
    }

    // ---- heads (net.py:62-136): whole wave, all P positions at once (shared FC weight loads); dead positions unstored
    {
        const int c = lane < CELLS ? lane : 0;
        const float* srcs[P];
        float* lps[P];
        float* vs[P];
        bool live[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            srcs[p] = act + p * POSW + (c / BS + 1) * PW + (c % BS + 1);
            lps[p] = logp + (pos0 + p) * NP;
            vs[p] = vout + pos0 + p;
            live[p] = pos0 + p < nv;
        }
        heads_wave_n<F, BS, P>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, PS, scratch, lane, lps, vs, live);
    }
}

// ------------------------------------------------------------------------------------------------
// host: pack weights into fragment order
// ------------------------------------------------------------------------------------------------
void f32_free_weights(oth_net* net) {
    if (!net->f32) return;
    if (net->f32->d_w) (void)hipFree(net->f32->d_w);
    if (net->f32->d_bias) (void)hipFree(net->f32->d_bias);
    if (net->f32->d_pfc_wt) (void)hipFree(net->f32->d_pfc_wt);
    if (net->f32->d_vfc1_wt) (void)hipFree(net->f32->d_vfc1_wt);
    delete net->f32;
    net->f32 = nullptr;
}

// FoldedConv weights are [tap][cin][cout]; lane l of k-step kc needs, for e = tap*NB + blk,
// W[cout = 16*blk + (l & 15)][cin = 4*kc + (l >> 4)][tap]  (cin >= c.cin: zero -- the stem's fourth plane)
static void pack_layer(const FoldedConv& c, int F, int kc_steps, std::vector<float>& out) {
    const int NB = F / 16, NW = 9 * NB, NCH = (NW + 3) / 4;
    const size_t base = out.size();
    out.resize(base + (size_t)kc_steps * NCH * 64 * 4, 0.f);
    for (int kc = 0; kc < kc_steps; ++kc)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < NW; ++e) {
                const int tap = e / NB, blk = e % NB;
                const int co = 16 * blk + (l & 15), ci = 4 * kc + (l >> 4);
                const float v = ci < c.cin ? c.w[((size_t)tap * c.cin + ci) * c.cout + co] : 0.f;
                out[base + (((size_t)kc * NCH + e / 4) * 64 + l) * 4 + (e % 4)] = v;
            }
}

int f32_pack_weights(oth_net* net) {
    const HostNet& hn = net->host;
    const int F = hn.filters, L = 1 + 2 * hn.blocks, cells = net->board * net->board, NP = cells + 1;
    OTH_CHECK(L <= kMaxTrunkLayers, "too many layers");
    F32Weights* fw = new F32Weights();
    net->f32 = fw;
    std::vector<float> w, bias((size_t)L * F);
    fw->layer_off.resize(L);
    for (int l = 0; l < L; ++l) {
        const FoldedConv& c = l == 0 ? hn.stem : hn.res[l - 1];
        fw->layer_off[l] = (uint32_t)(w.size() / 4);
        pack_layer(c, F, l == 0 ? 1 : F / 4, w);
        for (int i = 0; i < F; ++i) bias[(size_t)l * F + i] = c.bias[i];
    }
    std::vector<float> pt((size_t)2 * cells * NP), vt((size_t)cells * 256);
    for (int o = 0; o < NP; ++o)
        for (int i = 0; i < 2 * cells; ++i) pt[(size_t)i * NP + o] = hn.pfc_w[(size_t)o * 2 * cells + i];
    for (int o = 0; o < 256; ++o)
        for (int i = 0; i < cells; ++i) vt[(size_t)i * 256 + o] = hn.vfc1_w[(size_t)o * cells + i];
    OTH_HIP(hipMalloc(&fw->d_w, w.size() * 4));
    OTH_HIP(hipMalloc(&fw->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&fw->d_pfc_wt, pt.size() * 4));
    OTH_HIP(hipMalloc(&fw->d_vfc1_wt, vt.size() * 4));
    OTH_HIP(hipMemcpy(fw->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(fw->d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(fw->d_pfc_wt, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(fw->d_vfc1_wt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    return OTH_OK;
}

template <int F, int BS, int P, int WPB>
static int launch_f32(oth_net* net, const F32Args& a, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg,
                      int64_t n, const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    using G = TrunkGeom<F, BS, P>;
    constexpr size_t lds = (size_t)WPB * G::WAVE_WORDS * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set_dev[64] = {};  // per device: the attribute belongs to the (function, device) pair
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_f32<F, BS, P, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        attr_set = true;
    }
    const int64_t per_block = (int64_t)P * WPB;
    const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
    hipLaunchKernelGGL((k_trunk_f32<F, BS, P, WPB>), dim3(grid), dim3(64 * WPB), lds, stream, a, sb, ob, lg, n, n_valid,
                       logp, v);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

int f32_forward(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
                const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    OTH_CHECK(net->f32, "fp32 MFMA weights not packed");
    F32Args a;
    memset(&a, 0, sizeof(a));
    a.w = net->f32->d_w;
    a.bias = net->f32->d_bias;
    a.n_layers = 1 + 2 * net->blocks;
    for (int l = 0; l < a.n_layers; ++l) a.layer_off[l] = net->f32->layer_off[l];
    a.heads = net->heads;
    a.pfc_wt = net->f32->d_pfc_wt;
    a.vfc1_wt = net->f32->d_vfc1_wt;
    const int F = net->filters;
#define OTH_F32_CASE(FF, BB, PP, WW) \
    if (F == FF && net->board == BB) return launch_f32<FF, BB, PP, WW>(net, a, sb, ob, lg, n, n_valid, logp, v, stream)
    // positions per wave / waves per workgroup chosen for LDS (F planes of P positions per wave) and registers
    // (NB x T accumulators + as many residual registers per lane)
    OTH_F32_CASE(16, 8, 2, 4);
    OTH_F32_CASE(32, 8, 2, 4);
    OTH_F32_CASE(64, 8, 1, 4);
    OTH_F32_CASE(128, 8, 1, 2);
    OTH_F32_CASE(16, 6, 4, 4);
    OTH_F32_CASE(32, 6, 4, 4);
    OTH_F32_CASE(64, 6, 2, 4);
    OTH_F32_CASE(128, 6, 1, 4);
#undef OTH_F32_CASE
    set_error("fp32 trunk: unsupported filters %d / board %d", F, net->board);
    return OTH_E_UNSUPPORTED;
}

}  // namespace oth
