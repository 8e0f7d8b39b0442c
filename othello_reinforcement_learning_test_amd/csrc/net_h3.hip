// net_h3.hip -- OthelloResNet forward for 32 and 64 filters with the fp32-equivalent fp16 hi/lo operand split on
// v_mfma_f32_16x16x32_f16 (the arithmetic of net_mfma.hip's k_trunk16) in the barrier-free wave-per-position
// structure of net_f32.hip.  8x8 and 6x6 boards.
//
// Reference: /root/reference/src/model/net.py:139-205 (eval mode; BatchNorm folded at load time); 64 filters is
// configs/debug_6x6.yaml (5 blocks), 32 filters the width of tests/test_model.py-sized networks.
//
// Why a third trunk kernel: the exact-fp32 MFMA (net_f32.hip) peaks at 157 TFLOP/s, 1/16 of the f16 rate; three f16
// products per operand pair (a = a_hi + a_lo, a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in fp32) give the same ~22-bit
// operands at 3/16 of the cost.  k_trunk16 is built around 128 channels split over four waves; for narrow networks
// a whole position fits ONE wave:
//   * one wave carries P positions through stem, residual blocks and heads: no barrier anywhere;
//   * activations in LDS as f16 hi and lo arrays [8-channel chunk][cell][16 B]: the B operand of a tile (16
//     consecutive cells x 8 channels per k-group) is one ds_read_b128 per lane from 16 consecutive 16-byte slots per
//     k-group -- conflict-free without a swizzle.  Rows are NOT padded in x (a tile stays contiguous); the dx = +-1
//     taps read the neighbouring cell and the lanes whose source column is off the board zero their fragment
//     (v_cndmask); zero rows above and below each position make the dy = +-1 taps constant offsets;
//   * weights stream from L2 in A-fragment order (host-packed, 16 B per lane, one (tap, k-step) ahead in registers),
//     scaled per layer by a power of two so that the lo halves stay in the normal f16 range; activations and the fp32
//     residual are carried pre-scaled by 2^4; both scalings are undone exactly on the fp32 accumulator;
//   * a (tap, k-step) step feeds NB x T x 3 MFMAs (NB = F/16 row blocks, T tiles) from T activation and NB weight
//     fragment pairs; accumulators and residual stay in registers; the epilogue (bias, skip, ReLU, hi/lo re-split)
//     rewrites the activation arrays in place with 8-byte stores.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "net.h"
#include "net_heads_wave.h"
#include "net_epilogue.h"

namespace oth {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using half4 = _Float16 __attribute__((ext_vector_type(4)));
using f32x4 = float __attribute__((ext_vector_type(4)));


struct H3Weights {
    uint4* d_w = nullptr;        // [layer][tap][k-step][row block][hi, lo][64 lanes] x 16 B
    float* d_bias = nullptr;     // [layers][F], x act_scale (register_scaled_bias)
    float* d_inv = nullptr;      // [layers] 1 / weight scale
    float* d_pfc_wt = nullptr;   // transposed head FCs (net_heads_wave.h)
    float* d_vfc1_wt = nullptr;
    std::vector<uint32_t> layer_off;  // in uint4 units
};

struct H3Args {
    const uint4* w;
    uint32_t layer_off[kMaxTrunkLayers];
    const float* bias;
    const float* inv;
    int n_layers;
    HeadParams heads;
    const float* pfc_wt;
    const float* vfc1_wt;
    int* sat;   // set to 1 when an activation reaches the clamp (oth_net_saturated)
    float act_scale;   // oth_net::act_scale: the stem's input value and the heads' un-scaling
    unsigned long long* dbg;   // diagnostic build (-DOTH_STAMPS) only: per-wave phase cycle sums
};

template <int F, int BS, int P>
struct H3Geom {
    static constexpr int NB = F / 16;                 // 16-channel row blocks
    static constexpr int NKC = F / 8;                 // 8-channel chunks (16 B of f16)
    static constexpr int CELLS = BS * BS;
    static constexpr int NP = CELLS + 1;
    static constexpr int NCO = P * CELLS;             // output cells of a wave
    static constexpr int T = (NCO + 15) / 16;         // 16-cell tiles
    // Cell index of (position p, cell r): 1 + p*POSC + BS + r -- a guard cell, a zero row above, the board, and 16
    // zero cells (>= a zero row below plus the next position's zero row above) before the next position.  The 16-cell
    // gap and a chunk stride NC that is a multiple of 16 keep the 16-byte slot of (lane's cell, k-group) congruent to
    // `tile start + n16` modulo 16 for every lane of a ds_read_b128 group -- also in tiles that straddle two positions
    // -- so the reads are bank-conflict-free (with POSC = (BS+2)*BS and NC = P*POSC + 2, as in round 2, half of the LDS
    // cycles of this kernel were bank-conflict cycles: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50).
    static constexpr int POSC = CELLS + 16;
    static constexpr int NC = (P * POSC + 2 + 15) / 16 * 16;
    static_assert(2 * BS <= 16, "the inter-position gap must hold two zero rows");
    static constexpr int HI_BYTES = NKC * NC * 16;    // size of the hi array (= offset of the lo array)
    static constexpr int ACT_BYTES = 2 * HI_BYTES;
    static constexpr int SCRATCH_BYTES = 192 * 4 * P; // heads: 192 floats per position
    static constexpr int WAVE_BYTES = ACT_BYTES + SCRATCH_BYTES;
    static_assert(F * NCO * 4 <= ACT_BYTES, "the fp32 planes of the heads alias the activation arrays");
};

template <bool INPLACE>
__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) {
    if constexpr (INPLACE) {
        // accumulate in place (vDst = SrcC), as k_trunk16 does: no accumulator migration, no WAR wait states.  Inline
        // asm hides the instruction from hipcc's hazard recogniser: only for the instances tools/check_mfma_hazards.py
        // finds clean (the 32-filter builds get accumulator copies in front of their MFMAs: they use the builtin).
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        return c;
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
}

#ifdef OTH_STAMPS
__device__ __forceinline__ unsigned long long h3_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long h3_realclk() {   // 100 MHz constant clock
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define OTH_HSTAMP(i) { const unsigned long long t1_ = h3_clk(); ph_[i] += t1_ - t0_; t0_ = t1_; }
#else
#define OTH_HSTAMP(i)
#endif

// The WPB waves of a workgroup are independent (each streams its own weights from L2).  A build in which they shared the
// stream through a three-stage LDS ring (one barrier per step, OTH_H3_SHARE) was measured 5-7 % slower in round 3 -- the
// barrier and the ring's LDS round trip cost more than the quartered L2 -> CU stream saved -- and removed from the sources
// in round 5 (DESIGN_HISTORY.md K3c; history at 274f240).
template <int F, int BS, int P, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_trunk_h3(H3Args a, const uint64_t* __restrict__ sb,
                                                       const uint64_t* __restrict__ ob,
                                                       const uint64_t* __restrict__ lgl, int64_t n,
                                                       const int32_t* __restrict__ n_valid, float* __restrict__ logp,
                                                       float* __restrict__ vout) {
    using G = H3Geom<F, BS, P>;
    constexpr int NB = G::NB, CELLS = G::CELLS, NP = G::NP, NCO = G::NCO, T = G::T, POSC = G::POSC, NC = G::NC,
                  HI = G::HI_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds_h3[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
#ifdef OTH_STAMPS
    unsigned long long ph_[5] = {0, 0, 0, 0, 0}, t0_ = h3_clk();   // prologue | weight wait + conv | epilogue | planes | heads
    const unsigned long long tstart_ = t0_, rstart_ = h3_realclk();
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pos0 = ((int64_t)blockIdx.x * WPB + wave) * P;
    if (pos0 >= nv) return;   // the waves are independent
    const int n16 = lane & 15, g4 = lane >> 4;
    char* act = lds_h3 + (size_t)wave * G::WAVE_BYTES;
    float* scratch = (float*)(act + G::ACT_BYTES);

    // ---- zero both arrays (pad rows and guard cells stay zero for the whole network), then the input planes:
    //      channels 0..2 of chunk 0 = own / opponent / legal (bitboard.pyx:309-323) as exact f16 values 16.0 / 0
    for (int i = lane; i < G::ACT_BYTES / 16; i += 64) ((uint4*)act)[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (pos0 + p < nv && lane < CELLS) {
            const uint64_t b0 = sb[pos0 + p], b1 = ob[pos0 + p], b2 = lgl[pos0 + p];
            half4 v;
            v[0] = (b0 >> lane) & 1ULL ? (_Float16)a.act_scale : (_Float16)0.0f;
            v[1] = (b1 >> lane) & 1ULL ? (_Float16)a.act_scale : (_Float16)0.0f;
            v[2] = (b2 >> lane) & 1ULL ? (_Float16)a.act_scale : (_Float16)0.0f;
            v[3] = (_Float16)0.0f;
            const int idx = 1 + p * POSC + (lane / BS + 1) * BS + (lane % BS);
            *(half4*)(act + (size_t)idx * 16) = v;   // chunk 0, halfs 0..3
        }
    }

    // ---- per-lane tile geometry
    int rd_off[T], wr_off[T];   // byte offsets into the hi array
    bool valid[T], x_first[T], x_last[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ci = t * 16 + n16;
        valid[t] = ci < NCO;
        const int c = valid[t] ? ci : 0;
        const int p = c / CELLS, r = c % CELLS, x = r % BS;
        const int idx = 1 + p * POSC + BS + r;
        x_first[t] = x == 0;
        x_last[t] = x == BS - 1;
        rd_off[t] = (g4 * NC + idx) * 16;                           // B operand: chunk 4*kk + g4 of the source cell
        wr_off[t] = ((g4 >> 1) * NC + idx) * 16 + (g4 & 1) * 8;     // D rows 16*rb + 4*g4 + r: chunk 2*rb + (g4>>1)
    }
    const int zero_off = (g4 * NC) * 16;   // guard cell 0 of this lane's k-group chunk: always zero

    f32x4 acc[NB][T], res[NB][T];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            res[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            OTH_PIN_ACC(acc[b][t]);
        }
    OTH_PIN_ACC_END();

    uint32_t sat_bits = 0;   // largest activation seen (bit pattern); reaching the clamp raises the saturation flag
#define OTH_SB __builtin_amdgcn_sched_barrier(0)
    uint4 wq[NB * 2];          // the next step's fragments
    OTH_HSTAMP(0)
    for (int layer = 0; layer < a.n_layers; ++layer) {
        const int KK = layer == 0 ? 1 : F / 32;   // k-steps of 32 input channels (stem: planes 0..2 of chunk 0)
        const int nsteps = 9 * KK;
        const uint4* wl = a.w + a.layer_off[layer] + lane;   // + (step * NB * 2 + frag) * 64
        // bias and scale of this layer's epilogue, requested before the convolution: loaded inside the epilogue, each row
        // block waited ~700 cycles for its own L2 round trip with nothing else to run (one wave per SIMD) -- 2.8 k of
        // the epilogue's 3.8 k cycles per layer (in-kernel stamps)
        float4 bias_q[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bias_q[b] = *(const float4*)(a.bias + (size_t)layer * F + b * 16 + 4 * g4);
        const float inv = a.inv[layer];
#pragma unroll
        for (int f = 0; f < NB * 2; ++f) wq[f] = wl[(size_t)f * 64];
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                half8 wh[NB], wlo[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    wh[b] = __builtin_bit_cast(half8, wq[2 * b]);
                    wlo[b] = __builtin_bit_cast(half8, wq[2 * b + 1]);
                }
                int ns = kk * 9 + tap + 1;   // the next step's fragments (the last step re-reads its own: harmless)
                ns = ns < nsteps ? ns : nsteps - 1;
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int soff = (dy * BS + dx) * 16, koff = kk * 4 * NC * 16;
                // source address of tile t: lanes whose source column is off the board (first / last column of a row
                // for dx = -1 / +1) read the zero guard cell instead of zeroing eight fragment registers afterwards
                auto src_of = [&](int t) -> const char* {
                    int o = rd_off[t] + soff;
                    if (dx < 0) o = x_first[t] ? zero_off : o;
                    if (dx > 0) o = x_last[t] ? zero_off : o;
                    return act + o + koff;
                };
                // Issue order pinned by sched_barrier: the two LDS reads of tile t+1 and the weight loads of the next
                // step go BETWEEN the 3*NB MFMAs of tile t (one per MFMA slot) instead of in a clump in front of them,
                // where they left the matrix pipe idle (one wave per SIMD here: nothing else fills the gap).
                half8 xh[2], xl[2];
                xh[0] = *(const half8*)src_of(0);
                xl[0] = *(const half8*)(src_of(0) + HI);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int cur = t & 1, nxt = cur ^ 1;
                    const char* nsrc = t + 1 < T ? src_of(t + 1) : act;
                    OTH_SB;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {   // the three split products of one accumulator back to back
                        acc[b][t] = mfma_h<(F >= 64)>(wh[b], xl[cur], acc[b][t]);
                        OTH_SB;
                        if (b == 0 && t + 1 < T) xh[nxt] = *(const half8*)nsrc;
                        if (t == 0) wq[2 * b] = wl[((size_t)ns * NB * 2 + 2 * b) * 64];
                        OTH_SB;
                        acc[b][t] = mfma_h<(F >= 64)>(wh[b], xh[cur], acc[b][t]);
                        OTH_SB;
                        if (b == 0 && t + 1 < T) xl[nxt] = *(const half8*)(nsrc + HI);
                        if (t == 0) wq[2 * b + 1] = wl[((size_t)ns * NB * 2 + 2 * b + 1) * 64];
                        OTH_SB;
                        acc[b][t] = mfma_h<(F >= 64)>(wlo[b], xh[cur], acc[b][t]);
                        OTH_SB;
                    }
                }
            }
        }
        OTH_HSTAMP(1)
        // ---- epilogue: undo the weight scale, bias, skip connection (net.py:58-59), ReLU, re-split, rewrite in place.
        //      add_res / set_res are compile-time flags of two instantiations (no v_cndmask per value); the stem takes the
        //      skip form on a zero residual.
        auto epilogue = [&](auto ADD, auto SET) {
            constexpr bool add_res = decltype(ADD)::value;   // second conv of a block
            constexpr bool set_res = decltype(SET)::value;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float4 bias = bias_q[b];
                const f32x2 bias01 = {bias.x, bias.y}, bias23 = {bias.z, bias.w}, inv2 = {inv, inv};
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    // packed fp32 on the (0,1) / (2,3) channel pairs; the skip connection in place on the residual
                    // registers; the low parts straight from the packed hi (net_epilogue.h): 20 instead of ~31 VALU
                    // instructions per accumulator
                    f32x2 v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 tt = pk_fma(whalf(acc[b][t], h), inv2, h == 0 ? bias01 : bias23);
                        if (add_res) {
                            f32x2 r = whalf(res[b][t], h);
                            pk_add_relu_inplace(r, tt, 60000.f);   // ReLU + f16 range clamp
                            wsethalf(res[b][t], h, r);
                            v[h] = r;
                        } else {
                            v[h] = f32x2{__builtin_amdgcn_fmed3f(tt.x, 0.f, 60000.f), __builtin_amdgcn_fmed3f(tt.y, 0.f, 60000.f)};
                            if (set_res) wsethalf(res[b][t], h, v[h]);
                        }
                    }
                    sat_bits = max(sat_bits, max(max(__float_as_uint(v[0].x), __float_as_uint(v[0].y)),
                                                 max(__float_as_uint(v[1].x), __float_as_uint(v[1].y))));
                    if (valid[t]) {
                        uint2 hi, lo;
                        hi.x = wpack(v[0].x, v[0].y);
                        hi.y = wpack(v[1].x, v[1].y);
                        lo.x = wresid(hi.x, v[0].x, v[0].y);
                        lo.y = wresid(hi.y, v[1].x, v[1].y);
                        char* dst = act + wr_off[t] + b * 2 * NC * 16;
                        *(uint2*)dst = hi;
                        *(uint2*)(dst + HI) = lo;
                    }
                }
            }
        };
        // two variants: the stem goes through the skip form too (the residual registers hold zeros: t + 0 is exact)
        if (layer & 1) epilogue(std::false_type{}, std::false_type{});
        else epilogue(std::true_type{}, std::true_type{});
        // zero the accumulators for the next conv here (not in the epilogue, where they would hold registers), pinned
        // in front of the wait states the in-place asm MFMAs need after a VALU write (net.h)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                OTH_PIN_ACC(acc[b][t]);
            }
        OTH_PIN_ACC_END();
        OTH_HSTAMP(2)
    }
#undef OTH_SB

    // ---- heads: the final activations (in `res`, x act_scale) as fp32 planes [channel][output cell], aliasing
    //      the activation arrays (every read of them is done)
    if (sat_bits >= __float_as_uint(60000.f)) atomicOr(a.sat, 1);   // rare: surfaced by oth_net_saturated
    float* planes = (float*)act;
    const float us = 1.0f / a.act_scale;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (valid[t]) {
                const int ci = t * 16 + n16;
#pragma unroll
                for (int r = 0; r < 4; ++r) planes[(b * 16 + 4 * g4 + r) * NCO + ci] = res[b][t][r] * us;
            }
    OTH_HSTAMP(3)
    {   // all P positions of the wave at once (shared FC weight loads); dead positions compute on zero planes, unstored
        const int c = lane < CELLS ? lane : 0;
        const float* srcs[P];
        float* lps[P];
        float* vs[P];
        bool live[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            srcs[p] = planes + p * CELLS + c;
            lps[p] = logp + (pos0 + p) * NP;
            vs[p] = vout + pos0 + p;
            live[p] = pos0 + p < nv;
        }
        heads_wave_n<F, BS, P>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, NCO, scratch, lane, lps, vs, live);   // (WIDE batches measured here too: no change, 0.487 vs 0.490 ms for 5x64 on 8x8)
    }
#ifdef OTH_STAMPS
    OTH_HSTAMP(4)
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * WPB + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ph_[i];
        o[5] = h3_clk() - tstart_;
        o[6] = h3_realclk() - rstart_;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host: pack weights into A-fragment order
// ------------------------------------------------------------------------------------------------
void h3_free_weights(oth_net* net) {
    if (!net->h3) return;
    H3Weights* w = net->h3;
    if (w->d_w) (void)hipFree(w->d_w);
    if (w->d_bias) (void)hipFree(w->d_bias);
    if (w->d_inv) (void)hipFree(w->d_inv);
    if (w->d_pfc_wt) (void)hipFree(w->d_pfc_wt);
    if (w->d_vfc1_wt) (void)hipFree(w->d_vfc1_wt);
    delete w;
    net->h3 = nullptr;
}

static inline void split_h3(float v, uint16_t& hi, uint16_t& lo) {
    const _Float16 hh = (_Float16)v;
    const _Float16 ll = (_Float16)(v - (float)hh);
    memcpy(&hi, &hh, 2);
    memcpy(&lo, &ll, 2);
}

// A fragment of v_mfma_f32_16x16x32_f16: lane l holds W[row = l&15][k = 8*(l>>4) + j], j = 0..7.
// FoldedConv weights are [tap][cin][cout]; cin >= c.cin is zero (the stem's 3 planes in a 32-wide k-step).
static float pack_layer_h3(const FoldedConv& c, int F, int kk_steps, std::vector<uint16_t>& out) {
    const int NB = F / 16;
    float mx = 0.f;
    for (float x : c.w) mx = fmaxf(mx, fabsf(x));
    int e = 0;
    if (mx > 0.f) e = (int)floorf(log2f(16384.0f / mx));  // largest |w| lands in [8192, 16384]
    if (e > 24) e = 24;
    if (e < -24) e = -24;
    const float scale = ldexpf(1.0f, e);
    const size_t base = out.size();
    out.resize(base + (size_t)9 * kk_steps * NB * 2 * 64 * 8, 0);
    for (int kk = 0; kk < kk_steps; ++kk)
        for (int tap = 0; tap < 9; ++tap)
            for (int b = 0; b < NB; ++b)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 16 * b + (l & 15), ci = 32 * kk + 8 * (l >> 4) + j;
                        const float v = ci < c.cin ? c.w[((size_t)tap * c.cin + ci) * c.cout + co] * scale : 0.f;
                        uint16_t hi, lo;
                        split_h3(v, hi, lo);
                        const size_t step = (size_t)kk * 9 + tap;
                        const size_t frag = base + ((step * NB + b) * 2) * 64 * 8;
                        out[frag + (size_t)l * 8 + j] = hi;
                        out[frag + 64 * 8 + (size_t)l * 8 + j] = lo;
                    }
    return scale;
}

int h3_pack_weights(oth_net* net) {
    const HostNet& hn = net->host;
    const int F = hn.filters, L = 1 + 2 * hn.blocks, cells = net->board * net->board, NP = cells + 1;
    OTH_CHECK(F == 32 || F == 64 || (F == 128 && net->board == 6),
              "the wave-per-position fp16-split trunk is built for 32 and 64 filters (and 128 on 6x6)");
    OTH_CHECK(L <= kMaxTrunkLayers, "too many layers");
    H3Weights* hw = new H3Weights();
    net->h3 = hw;
    std::vector<uint16_t> w;
    std::vector<float> bias((size_t)L * F), inv(L);
    hw->layer_off.resize(L);
    for (int l = 0; l < L; ++l) {
        const FoldedConv& c = l == 0 ? hn.stem : hn.res[l - 1];
        hw->layer_off[l] = (uint32_t)(w.size() / 8);
        const float sc = pack_layer_h3(c, F, l == 0 ? 1 : F / 32, w);
        inv[l] = 1.0f / sc;   // accumulator = (16 x) * (sc w): times 1/sc gives 16 * y
        for (int i = 0; i < F; ++i) bias[(size_t)l * F + i] = c.bias[i];
    }
    std::vector<float> pt((size_t)2 * cells * NP), vt((size_t)cells * 256);
    for (int o = 0; o < NP; ++o)
        for (int i = 0; i < 2 * cells; ++i) pt[(size_t)i * NP + o] = hn.pfc_w[(size_t)o * 2 * cells + i];
    for (int o = 0; o < 256; ++o)
        for (int i = 0; i < cells; ++i) vt[(size_t)i * 256 + o] = hn.vfc1_w[(size_t)o * cells + i];
    OTH_HIP(hipMalloc(&hw->d_w, w.size() * 2));
    OTH_HIP(hipMalloc(&hw->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&hw->d_inv, inv.size() * 4));
    OTH_HIP(hipMalloc(&hw->d_pfc_wt, pt.size() * 4));
    OTH_HIP(hipMalloc(&hw->d_vfc1_wt, vt.size() * 4));
    OTH_HIP(hipMemcpy(hw->d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    if (int rc = register_scaled_bias(net, hw->d_bias, std::move(bias))) return rc;
    OTH_HIP(hipMemcpy(hw->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(hw->d_pfc_wt, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(hw->d_vfc1_wt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    return OTH_OK;
}

template <int F, int BS, int P, int WPB>
static int launch_h3(oth_net* net, const H3Args& a, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg,
                     int64_t n, const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    using G = H3Geom<F, BS, P>;
    constexpr size_t lds = (size_t)WPB * G::WAVE_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set_dev[64] = {};  // per device: the attribute belongs to the (function, device) pair
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_h3<F, BS, P, WPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        attr_set = true;
    }
    const int64_t per_block = (int64_t)P * WPB;
    const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
#ifdef OTH_STAMPS
    H3Args ad = a;
    OTH_HIP(hipMalloc(&ad.dbg, (size_t)grid * WPB * 8 * sizeof(unsigned long long)));
    OTH_HIP(hipMemset(ad.dbg, 0, (size_t)grid * WPB * 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((k_trunk_h3<F, BS, P, WPB>), dim3(grid), dim3(64 * WPB), lds, stream, ad, sb, ob, lg, n, n_valid,
                       logp, v);
    OTH_HIP(hipStreamSynchronize(stream));
    {
        std::vector<unsigned long long> h((size_t)grid * WPB * 8);
        OTH_HIP(hipMemcpy(h.data(), ad.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double sm[7] = {0, 0, 0, 0, 0, 0, 0};
        size_t nw = 0;
        for (size_t w = 0; w < (size_t)grid * WPB; ++w) {
            if (!h[w * 8 + 5]) continue;
            ++nw;
            for (int i = 0; i < 7; ++i) sm[i] += (double)h[w * 8 + i];
        }
        fprintf(stderr, "[h3 stamps %dx%d P%d] per-wave cycles: prologue %.0f | weight wait + conv %.0f | epilogues %.0f | planes %.0f | heads %.0f | total %.0f | clock %.3f GHz\n",
                F, BS, P, sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw, sm[5] / nw, sm[5] / sm[6] * 0.1);
        (void)hipFree(ad.dbg);
    }
    return OTH_OK;
#endif
    hipLaunchKernelGGL((k_trunk_h3<F, BS, P, WPB>), dim3(grid), dim3(64 * WPB), lds, stream, a, sb, ob, lg, n, n_valid,
                       logp, v);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

int h3_forward(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
               const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    OTH_CHECK(net->h3, "fp16-split weights not packed");
    H3Args a;
    memset(&a, 0, sizeof(a));
    a.w = net->h3->d_w;
    a.bias = net->h3->d_bias;
    a.inv = net->h3->d_inv;
    a.n_layers = 1 + 2 * net->blocks;
    for (int l = 0; l < a.n_layers; ++l) a.layer_off[l] = net->h3->layer_off[l];
    a.heads = net->heads;
    a.pfc_wt = net->h3->d_pfc_wt;
    a.vfc1_wt = net->h3->d_vfc1_wt;
    a.sat = net->d_sat;
    a.act_scale = net->act_scale;
    const int F = net->filters;
#define OTH_H3_CASE(FF, BB, PP, WW) \
    if (F == FF && net->board == BB) return launch_h3<FF, BB, PP, WW>(net, a, sb, ob, lg, n, n_valid, logp, v, stream)
    OTH_H3_CASE(64, 8, 1, 4);
    OTH_H3_CASE(32, 8, 1, 4);
    OTH_H3_CASE(64, 6, 2, 4);
    OTH_H3_CASE(32, 6, 4, 4);
    OTH_H3_CASE(128, 6, 1, 4);   // 128 filters on 6x6 (8x8 has k_trunk16)
#undef OTH_H3_CASE
    set_error("fp16-split wave trunk: unsupported filters %d / board %d", F, net->board);
    return OTH_E_UNSUPPORTED;
}

}  // namespace oth
