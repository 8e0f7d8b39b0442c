// net_wino.hip -- OthelloResNet forward for 128 filters on 8x8 with the residual 3x3 convolutions as a 1-D Winograd
// F(2,3) along the board's x axis (the three row taps stay direct): 4 multiplies per 2 outputs instead of 6, i.e.
// 1.5x fewer MFMAs than the direct form (1.375x fewer than k_trunk16, which skips the all-padding row tiles).
//
// Reference: /root/reference/src/model/net.py:182-205 (eval mode; BatchNorm folded at load time).  Same arithmetic
// contract as net_mfma.hip: both operands split a = a_hi + a_lo (two f16), three products accumulated in fp32.
// Numerics of the transform with that split, against float64 on the trained-like network: tools/winograd1d_numerics.py
// (F(2,3) 1.9e-5 / 8.0e-5 at 6 / 10 blocks vs 3.4e-5 / 6.6e-5 direct; F(4,3) fails the 1e-4 bar).
//
//   y(x) = sum_i g_i in(x + i - 1);  tile j = outputs x = 2j, 2j+1 from inputs d_k = in(2j - 1 + k), k = 0..3
//   U = G g : u0 = g0, u1 = (g0+g1+g2)/2, u2 = (g0-g1+g2)/2, u3 = g2          (host, float64, then split into two f16)
//   V = B'd : v0 = d0-d2, v1 = d1+d2, v2 = d2-d1, v3 = d1-d3                  (epilogue, fp32, then split into two f16)
//   M_xi = sum over (row tap dy, input channel) of U[dy][xi] V[xi](row y+dy);  y0 = M0+M1+M2, y1 = M1-M2-M3
//
// One 512-thread workgroup (8 waves) per CU carries TWO positions through the whole network.  Wave w owns output
// channels [16w, 16w+16) of both positions: 4 N-tiles of 16 Winograd tiles (position p, board half h: rows 4h..4h+3 x
// tile column j) x 4 transformed taps = 16 accumulators.  The MFMA result keeps the tile on the lane and the four xi in
// four accumulators, so the OUTPUT transform is in-lane; the INPUT transform of the next layer needs in(2j-1) and
// in(2j+2) from the neighbouring tile columns = the neighbouring lanes of a quad (DPP quad_perm), in registers too.  The
// transformed operand V lives in LDS as [position][tile 0..31][xi][hi 128 x f16 | lo 128 x f16] = 128 KB (the reason
// for one workgroup per CU), 16-byte chunks XOR-swizzled with the tile index so that every ds_read_b128 lane group is
// conflict-free; a row tap is a shift of 4 tiles; out-of-board rows read a zero block.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "net.h"
#include "net_heads.h"
#include "net_epilogue.h"

namespace oth {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using half4 = _Float16 __attribute__((ext_vector_type(4)));
using f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int kWTile = 512;                   // bytes of one (position, tile, xi): 256 hi + 256 lo
constexpr int kWVBytes = 2 * 32 * 4 * kWTile; // 131072
constexpr int kWZeroOff = kWVBytes;           // 4 x 512 B of zeros: the source of out-of-board rows, any xi
constexpr int kWLds = kWVBytes + 4 * kWTile;  // 133120
constexpr float kWClamp = 30000.0f;           // |V| <= 2 x (activation x act_scale) must stay in the f16 range: activations
                                              // <= 30000 / act_scale (1875 at the default scale 16; oth_net::act_scale)

struct WinoWeights {
    int blocks = 0;
    uint4* d_w = nullptr;     // [layer][dy 3][kk 4][wave 8][xi 4][hi, lo][64 lanes] x 16 B   (A fragments of U)
    uint4* d_stem = nullptr;  // [wave 8][hi, lo][64 lanes] x 16 B: direct 3x3 stem as one k-step of 32 (27 used)
    float* d_bias = nullptr;  // [1 + 2*blocks][128], x act_scale (register_scaled_bias)
    float* d_inv = nullptr;   // [1 + 2*blocks] 1 / weight scale
    float* d_pfc_wt = nullptr;   // [128][65]  policy FC transposed: lanes read consecutive outputs
    float* d_vfc1_wt = nullptr;  // [64][256]  value FC1 transposed
};

struct WinoArgs {
    const uint4* w;
    const uint4* stem;
    const float* bias;
    const float* inv;
    int n_res_layers;
    HeadParams heads;
    const float* pfc_wt;
    const float* vfc1_wt;
    int* sat;
    float act_scale;           // oth_net::act_scale: the stem's input value and the heads' un-scaling
    unsigned long long* dbg;   // diagnostic build (-DOTH_STAMPS) only: per-wave phase cycle sums
};

#ifdef OTH_STAMPS
__device__ __forceinline__ unsigned long long w_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long w_realclk() {   // 100 MHz constant clock
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define OTH_WSTAMP(i) { const unsigned long long t1_ = w_clk(); ph_[i] += t1_ - t0_; t0_ = t1_; }
#else
#define OTH_WSTAMP(i)
#endif

__device__ __forceinline__ f32x4 wmfma(half8 a, half8 b, f32x4 c) {
    // in place (vDst = SrcC); tools/check_mfma_hazards.py checks the built object (see net.h)
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x4 wmfma0(half8 a, half8 b) {   // first product of a chain: C is the inline constant 0
    f32x4 c;
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ void wbarrier() {   // LDS-only barrier: global weight prefetches stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// value of the previous / next lane of this lane's quad (tile column j -+ 1); lanes at the ends get their own value
__device__ __forceinline__ float quad_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x90, 0xf, 0xf, true));  // [0,0,1,2]
}
__device__ __forceinline__ float quad_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xF9, 0xf, 0xf, true));  // [1,2,3,3]
}

#define OTH_WSB __builtin_amdgcn_sched_barrier(0)
constexpr int kHeadRow = 129;   // floats per cell of the heads' fp32 planes (odd: conflict-free column reads)

// Policy and value heads of BOTH positions by the whole 512-thread workgroup (net.py:83-96, 119-136; fp32 VALU).  With one
// workgroup per CU nothing overlaps the heads, and the shared heads_forward (one position at a time, 256 threads, FC
// rows read with a 256-byte stride per lane) took ~50 k cycles per position there -- a fifth of this kernel.  Here:
// the 1x1-conv weights staged in LDS, the FC weights transposed on the host so that lanes read consecutive outputs,
// both positions at once.  act: fp32 [128 cells][kHeadRow = 129 floats: 128 channels + 1 pad]; scratch: 1 568 floats of LDS.
__device__ __forceinline__ void heads_block2(const HeadParams& hp, const float* __restrict__ pfc_wt,
                                             const float* __restrict__ vfc1_wt, const float* act, float* scratch,
                                             bool live1, float* __restrict__ logp, float* __restrict__ vout,
                                             unsigned long long* hst = nullptr) {
    const int t = threadIdx.x;
#ifdef OTH_STAMPS
    unsigned long long h0_ = w_clk();
#define OTH_HST(i) { const unsigned long long h1_ = w_clk(); if (hst) hst[i] = h1_ - h0_; h0_ = h1_; }
#else
#define OTH_HST(i)
#endif
    float* w3 = scratch;             // [128][3]: policy conv 0, policy conv 1, value conv
    float* feat = scratch + 384;     // [2 positions][192]: policy features (channel, cell) then value features
    float* h1 = scratch + 768;       // [2][256]
    float* lgp = scratch + 1280;     // [2][2][72]: partial logits of the policy FC (two input halves)
    if (t < 384) {
        const int i = t / 3, k = t % 3;
        w3[t] = k < 2 ? hp.pconv_w[i * 2 + k] : hp.vconv_w[i];
    }
    __syncthreads();
    if (t < 384) {
        const int p = t / 192, rem = t % 192, k = rem >> 6, cell = rem & 63;
        const float* a = act + (size_t)(p * 64 + cell) * kHeadRow;
        // rows of 129 floats: lane = cell reads a[i] at bank (cell + i) mod 32 -- conflict-free WITHOUT a per-lane
        // rotation, so both LDS reads of an iteration have immediate offsets (the weight read is a broadcast) and an
        // iteration is 2 reads + 1 fma instead of ~10 instructions of address arithmetic.  (The weights as scalar loads
        // from global memory instead: slower, 3.7 k vs 2.8 k cycles -- scalar-cache misses.)
        float p0 = 0.f, p1 = 0.f;
#pragma unroll 16
        for (int i = 0; i < 128; i += 2) {
            p0 = fmaf(a[i], w3[i * 3 + k], p0);
            p1 = fmaf(a[i + 1], w3[(i + 1) * 3 + k], p1);
        }
        float acc = p0 + p1;
        acc += k < 2 ? hp.pconv_b[k] : hp.vconv_b[0];
        feat[p * 192 + k * 64 + cell] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    OTH_HST(0)
    // both FCs read L2-resident weights (97 KB): every weight is loaded ONCE and used for both positions, the loops are
    // fully unrolled (64 independent loads in flight per thread); threads 0..255 do the value FC1 (one output each),
    // threads 256..385 the policy FC (65 outputs x two halves of the 128 inputs)
    if (t < 256) {
        float a0 = hp.vfc1_b[t], a1 = a0;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float w = vfc1_wt[i * 256 + t];
            a0 = fmaf(w, feat[128 + i], a0);
            a1 = fmaf(w, feat[192 + 128 + i], a1);
        }
        h1[t] = a0 > 0.f ? a0 : 0.f;
        h1[256 + t] = a1 > 0.f ? a1 : 0.f;
    } else if (t < 256 + 130) {
        const int u = t - 256, o = u % 65, part = u / 65;
        float s0 = part == 0 ? hp.pfc_b[o] : 0.f, s1 = s0;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float w = pfc_wt[(part * 64 + i) * 65 + o];
            s0 = fmaf(w, feat[part * 64 + i], s0);
            s1 = fmaf(w, feat[192 + part * 64 + i], s1);
        }
        lgp[part * 72 + o] = s0;
        lgp[(2 + part) * 72 + o] = s1;
    }
    OTH_HST(1)
    __syncthreads();
    if ((t & 255) < 64) {   // waves 0 and 4: log_softmax over the 65 logits and the fc2 dot product of their position
        const int p = t >> 8, l = t & 63;
        const float* gp = lgp + p * 2 * 72;
        const float gl = gp[l] + gp[72 + l];
        const float g64 = gp[64] + gp[72 + 64];
        float m = fmaxf(gl, l == 0 ? g64 : -INFINITY);
        m = bfly_max_f32(m);
        float s = expf(gl - m) + (l == 0 ? expf(g64 - m) : 0.f);
        s = bfly_sum_f32(s);
        const float lse = logf(s);
        float acc = 0.f;
        for (int i = l; i < 256; i += 64) acc = fmaf(hp.vfc2_w[i], h1[p * 256 + i], acc);
        acc = bfly_sum_f32(acc);
        if (p == 0 || live1) {
            logp[p * 65 + l] = gl - m - lse;
            if (l == 0) {
                logp[p * 65 + 64] = g64 - m - lse;
                vout[p] = tanhf(acc + hp.vfc2_b[0]);
            }
        }
    }
}

// TP = 2: two positions per workgroup (4 N-tiles), the throughput build.  TP = 1: one position (2 N-tiles), the
// low-latency build for launches of <= 256 positions (every workgroup has a CU to itself there): eight waves still
// split the 128 channels, so a position's layer is 288 MFMAs per wave instead of the direct kernel's 792 on four.
// Both builds sum every accumulator in the same order: a position's outputs do not depend on the launch size.
template <int TP>
__global__ __launch_bounds__(512, 2) void k_trunk_w(WinoArgs a, const uint64_t* __restrict__ sb,
                                                    const uint64_t* __restrict__ ob, const uint64_t* __restrict__ lgl,
                                                    int64_t n, const int32_t* __restrict__ n_valid,
                                                    float* __restrict__ logp, float* __restrict__ vout) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef OTH_STAMPS
    const unsigned long long tentry_ = w_clk();
#endif
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    constexpr int NT = 2 * TP;   // N-tiles of a wave: (position, board half)
    const int64_t pos0 = (int64_t)blockIdx.x * TP;   // a workgroup takes position group blockIdx.x
    if (pos0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, c = lane & 15;
    const int row4 = c >> 2, j = c & 3;        // row within the board half, tile column
    // N-tile nt = 2*p + h: position p, board half h; this lane's tile index within the position: 16*h + c

    // ---- stem input: im2col of the three bit planes, [128 cells][32 k] f16 (64 B per cell) at LDS 0 (V is not live yet)
    if (tid < TP * 64) {
        const int p = tid >> 6, cell = tid & 63, y = cell >> 3, x = cell & 7;
        const bool live = pos0 + p < nv;
        const uint64_t b0 = live ? sb[pos0 + p] : 0, b1 = live ? ob[pos0 + p] : 0, b2 = live ? lgl[pos0 + p] : 0;
        // the rows of odd-x output cells are NEGATED: the stem's result then has the form of a Winograd-domain
        // accumulator set (M0 = y0, M1 = M2 = 0, M3 = -y1) and goes through the same epilogue as every other layer
        const _Float16 one = (x & 1) ? (_Float16)(-a.act_scale) : (_Float16)a.act_scale;
        _Float16 vals[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) vals[i] = (_Float16)0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
            const int s = ok ? yy * 8 + xx : 0;
            vals[tap * 3 + 0] = (ok && ((b0 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 1] = (ok && ((b1 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 2] = (ok && ((b2 >> s) & 1ULL)) ? one : (_Float16)0.0f;
        }
        half8* dst = (half8*)(lds + tid * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = vals[q * 8 + i];
            dst[q] = t;
        }
    }
    if (tid < 128) ((uint4*)(lds + kWZeroOff))[tid] = make_uint4(0, 0, 0, 0);   // 2 KB of zeros
    __syncthreads();

    f32x4 acc[4][NT];    // [xi][N-tile]
    f32x4 res[NT][2];    // [N-tile][x parity]: the residual in the spatial domain, fp32, x act_scale
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[xi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            OTH_PIN_ACC(acc[xi][nt]);
        }
    OTH_PIN_ACC_END();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) res[nt][0] = res[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f};   // the stem "adds" to this

    {   // ---- stem conv (net.py:195), direct: for each N-tile the even-x and the odd-x cells are two column sets:
        //      acc[0] = y0 and acc[3] = -y1 of the Winograd tile (acc[1] = acc[2] = 0), so that the output transform
        //      y0 = M0 + M1 + M2, y1 = M1 - M2 - M3 of the common epilogue returns them exactly
        const uint4* wp = a.stem + (size_t)wave * 2 * 64 + lane;
        const half8 wh = __builtin_bit_cast(half8, wp[0]), wlo = __builtin_bit_cast(half8, wp[64]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cell = (nt >> 1) * 64 + ((nt & 1) * 4 + row4) * 8 + 2 * j + e;
                const half8 xh = *(const half8*)(lds + cell * 64 + g4 * 16);
                acc[3 * e][nt] = wmfma(wlo, xh, acc[3 * e][nt]);
                acc[3 * e][nt] = wmfma(wh, xh, acc[3 * e][nt]);
            }
    }

    // lane constants of the V layout
    const int ch0 = wave * 16 + 4 * g4;                                   // + r: this lane's four channels
    const uint32_t wchunk = (uint32_t)(2 * wave + (g4 >> 1));            // 16-byte chunk of those channels
    uint32_t wr_off[NT];                                                  // store address of (N-tile, xi = 0), hi part
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const uint32_t tl = (uint32_t)((nt & 1) * 16 + c);
        wr_off[nt] = ((uint32_t)(nt >> 1) * 32 + tl) * (4 * kWTile) + ((wchunk ^ (tl & 15)) << 4) + 8u * (uint32_t)(g4 & 1);
    }
    uint32_t rd_base[NT][3];  // read base of (N-tile, row tap): the tile 4*dy further, or the zero block
    uint32_t rd_key[3];       // (source tile & 15) << 4, xor-ed with the k-group chunk
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        rd_key[d] = ((uint32_t)((c + 4 * (d - 1)) & 15) << 4) ^ ((uint32_t)g4 << 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int y = (nt & 1) * 4 + row4 + (d - 1);
            const int ts = (nt & 1) * 16 + c + 4 * (d - 1);
            rd_base[nt][d] = (y >= 0 && y < 8) ? (uint32_t)((nt >> 1) * 32 + ts) * (4 * kWTile) : (uint32_t)kWZeroOff;
        }
    }
    const float mask_l = j == 0 ? 0.f : 1.f, mask_r = j == 3 ? 0.f : 1.f;
    const f32x2 mask_l2 = {mask_l, mask_l}, mask_r2 = {mask_r, mask_r};

    const int n_layers = 1 + a.n_res_layers;
    uint32_t sat_bits = 0;
    // weight ring: [group mod RING][xi hi, xi lo]; a group = (row tap, k-step): 8 fragments.  Two groups at TP = 2; FOUR
    // at TP = 1, where the launch is one position per CU and bound by how many bytes of the weight stream (15.7 MB per
    // forward through ONE CU's vector-memory path), not by MFMAs: one position 0.177 -> 0.172 ms with 3 or 4 groups in flight
    // (6 spills: 0.259); the same loads with the non-temporal hint: 0.258 ms (they lose their L2 hits)
    constexpr int RING = TP == 1 ? 4 : 2;
    static_assert(12 % RING == 0, "the ring index must be a compile-time function of the group");
    uint4 wq[RING][8];
#ifdef OTH_STAMPS
    unsigned long long ph_[4] = {0, 0, 0, 0}, t0_ = w_clk(), tstart_ = t0_;
    const unsigned long long rstart_ = w_realclk();
#endif
    float4 b4 = *(const float4*)(a.bias + ch0), b4n = b4;   // bias (x act_scale) and 1 / weight scale of the layer in the epilogue
    float inv = a.inv[0], invn = inv;
    for (int layer = 0; layer < n_layers; ++layer) {
        const bool last = layer == n_layers - 1;
        if (layer > 0) {   // loaded during the previous convolution (at the loop top their L2 latency sat in front of the epilogue)
            b4 = b4n;
            inv = invn;
        }
        const f32x2 inv2 = {inv, inv};
        // A fragments of conv `layer+1`: group g = dy*4 + kk at wl + g * (8 waves * 8 frags * 64) uint4
        const uint4* wl = a.w + (size_t)layer * (12 * 8 * 8 * 64) + (size_t)wave * (8 * 64) + lane;
        // ---------------- epilogue of conv `layer`: output transform, scale, bias, skip, ReLU; then the next layer's
        //                  input transform and the hi/lo re-split into V
        // Two variants only -- SKIP (even layers: the stem, where the residual registers hold zeros, and the second
        // convolution of every block: add the residual and keep the result as the new one, IN PLACE) and plain (the
        // first convolution of a block) -- and the last layer, always a SKIP one, branches around the V part at run
        // time.  With five compile-time variants in the loop hipcc carried the residual through three register sets
        // (64 v_mov per even layer).
        auto epilogue = [&](auto SKIP) {
            constexpr bool add_res = decltype(SKIP)::value;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // packed fp32 throughout (v_pk_add/fma_f32 on the (r, r+1) register pairs; left to hipcc about half of
                // this was scalarised): 66 instead of 77 VALU instructions per N-tile, the same operation order
                f32x2 v0[2], v1[2];   // outputs x = 2j, 2j+1 of the tile, channel pairs (0,1) and (2,3)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 a0 = whalf(acc[0][nt], h), a1 = whalf(acc[1][nt], h), a2 = whalf(acc[2][nt], h),
                                a3 = whalf(acc[3][nt], h);
                    const f32x2 bb = h == 0 ? f32x2{b4.x, b4.y} : f32x2{b4.z, b4.w};
                    const f32x2 t0 = pk_fma(pk_add(pk_add(a0, a1), a2), inv2, bb);
                    const f32x2 t1 = pk_fma(pk_sub(pk_sub(a1, a2), a3), inv2, bb);
                    if (add_res) {
                        // skip connection IN PLACE on the residual registers: left to the allocator, the new residual
                        // went to a second register set and every layer paid 32-64 v_mov for the loop-carried values
                        f32x2 r0 = whalf(res[nt][0], h), r1 = whalf(res[nt][1], h);
                        pk_add_relu_inplace(r0, t0, kWClamp);
                        pk_add_relu_inplace(r1, t1, kWClamp);
                        wsethalf(res[nt][0], h, r0);
                        wsethalf(res[nt][1], h, r1);
                        v0[h] = r0;
                        v1[h] = r1;
                    } else {
                        v0[h] = f32x2{__builtin_amdgcn_fmed3f(t0.x, 0.f, kWClamp), __builtin_amdgcn_fmed3f(t0.y, 0.f, kWClamp)};
                        v1[h] = f32x2{__builtin_amdgcn_fmed3f(t1.x, 0.f, kWClamp), __builtin_amdgcn_fmed3f(t1.y, 0.f, kWClamp)};
                    }
                }
                sat_bits = max(sat_bits, max(max(__float_as_uint(v0[0].x), __float_as_uint(v0[0].y)),
                                             max(__float_as_uint(v0[1].x), __float_as_uint(v0[1].y))));
                sat_bits = max(sat_bits, max(max(__float_as_uint(v1[0].x), __float_as_uint(v1[0].y)),
                                             max(__float_as_uint(v1[1].x), __float_as_uint(v1[1].y))));
                if (!last) {
                    f32x2 V[4][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 dl = {quad_prev(v1[h].x), quad_prev(v1[h].y)};   // in(2j-1)
                        const f32x2 dr = {quad_next(v0[h].x), quad_next(v0[h].y)};   // in(2j+2)
                        V[0][h] = pk_fma_nc(dl, mask_l2, v1[h]);     // dl * mask - v1; the board's edge: in(-1) = in(8) = 0
                        V[1][h] = pk_add(v0[h], v1[h]);
                        V[2][h] = pk_sub(v1[h], v0[h]);
                        V[3][h] = pk_fma_na(dr, mask_r2, v0[h]);     // v0 - dr * mask
                    }
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
                        uint2 hi, lo;
                        hi.x = wpack(V[xi][0].x, V[xi][0].y);
                        hi.y = wpack(V[xi][1].x, V[xi][1].y);
                        lo.x = wresid(hi.x, V[xi][0].x, V[xi][0].y);
                        lo.y = wresid(hi.y, V[xi][1].x, V[xi][1].y);
                        char* dst = lds + wr_off[nt] + xi * kWTile;
                        *(uint2*)dst = hi;
                        *(uint2*)(dst + 256) = lo;
                    }
                }
                OTH_WSB;   // one N-tile at a time
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        // (Tried: the arithmetic of the epilogue BEFORE this barrier, in registers, so that the older wave of a SIMD --
        // which finishes its convolution at about half time under the oldest-first arbitration -- does it while the
        // younger one still convolves: 3.68 vs 2.69 ms.  A conv step already fills the SIMD's issue port (3 MFMAs x 8
        // cycles + 2 LDS reads + address VALU ~ 48 of its 48 cycles), so the extra VALU is not hidden, it slows the
        // partner's convolution.  Tried again with the arithmetic at s_setprio 0 and the convolution at s_setprio 3, whole
        // and for one or two N-tiles only: 3.60 / 2.94 / 3.04 ms -- the 16-64 extra live VGPRs spill at the 256 limit.)
        if (layer == 0 && !last) {   // the first weight group of the first convolution; the later ones load theirs in
                                     // the previous convolution's last group
#pragma unroll
            for (int g = 0; g < RING - 1; ++g)
#pragma unroll
                for (int f = 0; f < 8; ++f) wq[g][f] = wl[(size_t)g * (8 * 8 * 64) + (size_t)f * 64];
        }
        OTH_WSTAMP(0)
        wbarrier();   // every wave has finished reading V (or the stem's im2col)
        OTH_WSTAMP(1)
        if (layer & 1) epilogue(F_{});
        else epilogue(T_{});
        if (last) break;
        OTH_WSTAMP(2)
        wbarrier();
        OTH_WSTAMP(1)

        // ---------------- conv `layer+1` in the Winograd domain: 12 groups (row tap d, k-step kk) x 16 steps (N-tile,
        // xi) x 3 split products.  One straight-line software pipeline over all 192 steps: the two LDS reads of step
        // q+2 and, spread over a group, the eight weight loads of the next group sit between the MFMAs.
        // (Tried with the 30 registers the leaner epilogue freed: LDS prefetch distance 3 and 4, and the two waves of a SIMD
        // taking turns at s_setprio 2 group by group, to even out the oldest-first arbitration that lets one of them
        // finish its convolution ~8 k cycles before the other: 2.561-2.578 vs 2.562 ms -- no change.)
        // step q = ((d*4 + kk)*4 + nt)*4 + xi.  (Tried: xi as the outer index of a group with ONE 8-fragment weight ring --
        // the pair of (group, xi) reloaded right after its fourth use -- and per-use address arithmetic instead of the
        // hoisted (key ^ k-step) values: 32 VGPRs fewer, no spill, but 2.88 vs 2.69 ms: the extra VALU per step costs more
        // issue slots than the registers were worth.)
        constexpr int GS = NT * 4;          // steps of a group: (N-tile, xi)
        constexpr int QT = 12 * GS;         // steps of a layer
        auto src_of = [&](int q) -> const char* {
            const int xi = q & 3, nt = (q >> 2) % NT, grp = q / GS, kk = grp & 3, d = grp >> 2;
            return lds + rd_base[nt][d] + ((uint32_t)(kk << 6) ^ rd_key[d]) + xi * kWTile;
        };
        constexpr int PD = 2;   // LDS operand pairs in flight ahead of the MFMAs (steps)
        half8 xh[PD + 1], xl[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            xh[q] = *(const half8*)src_of(q);
            xl[q] = *(const half8*)(src_of(q) + 256);
        }
        // one instantiation per row tap (4 groups of straight-line code each, like k_trunk16's tap rows): a single loop
        // over the whole layer is beyond what hipcc unrolls, and a rolled loop turns every register array into
        // v_cndmask / v_readlane select chains
        auto conv_d = [&](auto DC) {
            constexpr int D = decltype(DC)::value;
#pragma unroll
            for (int ql = 0; ql < 4 * GS; ++ql) {
                const int q = D * 4 * GS + ql;
                const int xi = q & 3, nt = (q >> 2) % NT, grp = q / GS, sl = q % (PD + 1), psl = (q + PD) % (PD + 1);
                const int step = q % GS;                       // within the group
                const half8 wh = __builtin_bit_cast(half8, wq[grp % RING][2 * xi]);
                const half8 wlo = __builtin_bit_cast(half8, wq[grp % RING][2 * xi + 1]);
                OTH_WSB;
                if (q < GS) acc[xi][nt] = wmfma0(wh, xl[sl]);     // the layer's first group starts every accumulator
                else acc[xi][nt] = wmfma(wh, xl[sl], acc[xi][nt]);
                OTH_WSB;
                if (q + PD < QT) xh[psl] = *(const half8*)src_of(q + PD);
                OTH_WSB;
                acc[xi][nt] = wmfma(wh, xh[sl], acc[xi][nt]);
                OTH_WSB;
                if (q + PD < QT) xl[psl] = *(const half8*)(src_of(q + PD) + 256);
                // next group's fragments, one per step, as early as the ring allows (its other half is free from the
                // group's first step on): at TP = 2 that is >= 8 steps = 770+ cycles of cover for the L2 latency (issued
                // at steps 4..11 both waves of a SIMD stalled ~300 cycles at every group boundary)
                // (grp = 11 loads group 0 of the NEXT convolution -- the layers are contiguous, and one zero group pads the
                // end of the array for the last one: issued together before the barrier instead, the eight waves' 64 KB
                // burst through the CU's one 64 B/clk vector-memory path held every wave for ~1.1 k cycles per layer)
                if (grp == 10 && step == GS / 2) {
                    b4n = *(const float4*)(a.bias + (layer + 1) * 128 + ch0);
                    invn = a.inv[layer + 1];
                }
                if (step < 8)
                    wq[(grp + RING - 1) % RING][step] = wl[(size_t)(grp + RING - 1) * (8 * 8 * 64) + (size_t)step * 64];
                OTH_WSB;
                acc[xi][nt] = wmfma(wlo, xh[sl], acc[xi][nt]);
                OTH_WSB;
            }
        };
        conv_d(std::integral_constant<int, 0>{});
        conv_d(std::integral_constant<int, 1>{});
        conv_d(std::integral_constant<int, 2>{});
        OTH_WSTAMP(3)
    }
#ifdef OTH_STAMPS
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 4; ++i) o[i] = ph_[i];
        o[4] = w_clk() - tstart_;
        o[5] = w_realclk() - rstart_;
        o[6] = rstart_;
        o[7] = tstart_ - tentry_;   // prologue + stem
    }
    const unsigned long long theads_ = w_clk();
#endif

    // ---------------- heads (fp32 VALU): final activations (in `res`, x act_scale) -> LDS [128 cells][128] f32, then the
    // shared head code (256 of the 512 threads do the per-thread parts; all of them take its barriers)
    if (sat_bits >= __float_as_uint(kWClamp)) atomicOr(a.sat, 1);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cell = (nt >> 1) * 64 + ((nt & 1) * 4 + row4) * 8 + 2 * j + e;
            const float us = 1.0f / a.act_scale;
            const f32x4 v = res[nt][e];
            float* dst = (float*)lds + (size_t)cell * kHeadRow + ch0;   // odd row stride: four scalar stores
            dst[0] = v[0] * us;
            dst[1] = v[1] * us;
            dst[2] = v[2] * us;
            dst[3] = v[3] * us;
        }
    __syncthreads();
#ifdef OTH_STAMPS
    unsigned long long hst_[2] = {0, 0};
    const unsigned long long tpre_ = w_clk() - theads_;   // res -> LDS planes + barriers
    heads_block2(a.heads, a.pfc_wt, a.vfc1_wt, (const float*)lds, (float*)(lds + 128 * kHeadRow * 4 + 64), TP == 2 && pos0 + 1 < nv,
                 logp + pos0 * 65, vout + pos0, hst_);
#if OTH_STAMPS != 2   // -DOTH_STAMPS=2 keeps the layer phases (weight prefetch | barriers | epilogue | conv) instead
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[3] = w_clk() - theads_;   // (overwrites the conv sum)
        o[0] = tpre_; o[1] = hst_[0]; o[2] = hst_[1];   // (overwrite prefetch / barriers / epilogue sums)
    }
#endif
#else
    heads_block2(a.heads, a.pfc_wt, a.vfc1_wt, (const float*)lds, (float*)(lds + 128 * kHeadRow * 4 + 64), TP == 2 && pos0 + 1 < nv,
                 logp + pos0 * 65, vout + pos0);
#endif
}

// ------------------------------------------------------------------------------------------------
// host: transformed weights in A-fragment order
// ------------------------------------------------------------------------------------------------
static inline void wsplit(float v, uint16_t& hi, uint16_t& lo) {
    const _Float16 hh = (_Float16)v;
    const _Float16 ll = (_Float16)(v - (float)hh);
    memcpy(&hi, &hh, 2);
    memcpy(&lo, &ll, 2);
}

void wino_free_weights(oth_net* net) {
    if (!net->wino) return;
    WinoWeights* w = net->wino;
    if (w->d_w) (void)hipFree(w->d_w);
    if (w->d_stem) (void)hipFree(w->d_stem);
    if (w->d_bias) (void)hipFree(w->d_bias);
    if (w->d_inv) (void)hipFree(w->d_inv);
    if (w->d_pfc_wt) (void)hipFree(w->d_pfc_wt);
    if (w->d_vfc1_wt) (void)hipFree(w->d_vfc1_wt);
    delete w;
    net->wino = nullptr;
}

int wino_pack_weights(oth_net* net) {
    const HostNet& hn = net->host;
    OTH_CHECK(hn.filters == 128 && hn.board == 8, "the Winograd trunk is built for 128 filters on 8x8");
    const int L = 2 * hn.blocks;
    WinoWeights* ww = new WinoWeights();
    ww->blocks = hn.blocks;
    net->wino = ww;
    const size_t frag = 64 * 8;                              // halfs per fragment
    const size_t layer_halfs = (size_t)12 * 8 * 8 * frag;    // 12 groups x 8 waves x (4 xi x hi/lo)
    std::vector<uint16_t> w((size_t)L * layer_halfs + 5 * 8 * 8 * frag), stem((size_t)8 * 2 * frag);   // + zero groups: the last conv's look-ahead (ring depth - 1, at most 5)
    std::vector<float> bias((size_t)(L + 1) * 128), inv(L + 1);
    {   // stem: direct, gemm k = tap*3 + plane (27 of 32), rows = 16 channels of a wave
        const FoldedConv& cv = hn.stem;
        float mx = 0.f;
        for (float x : cv.w) mx = fmaxf(mx, fabsf(x));
        int e = mx > 0.f ? (int)floorf(log2f(16384.0f / mx)) : 0;
        e = e > 24 ? 24 : (e < -24 ? -24 : e);
        const float scale = ldexpf(1.0f, e);
        for (int wv = 0; wv < 8; ++wv)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int k = 8 * (l >> 4) + jj, co = 16 * wv + (l & 15);
                    const float v = k < 27 ? cv.w[(size_t)k * cv.cout + co] * scale : 0.f;   // [tap][cin=3][cout]: k = tap*3 + plane
                    uint16_t hi, lo;
                    wsplit(v, hi, lo);
                    stem[((size_t)wv * 2 + 0) * frag + (size_t)l * 8 + jj] = hi;
                    stem[((size_t)wv * 2 + 1) * frag + (size_t)l * 8 + jj] = lo;
                }
        inv[0] = 1.0f / scale;
        for (int i = 0; i < 128; ++i) bias[i] = cv.bias[i];
    }
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    std::vector<double> U((size_t)3 * 4 * 128 * 128);   // [dy][xi][ci][co]
    for (int li = 0; li < L; ++li) {
        const FoldedConv& cv = hn.res[li];
        double mx = 0.0;
        for (int d = 0; d < 3; ++d)
            for (int xi = 0; xi < 4; ++xi)
                for (int ci = 0; ci < 128; ++ci)
                    for (int co = 0; co < 128; ++co) {
                        double u = 0.0;
                        for (int i = 0; i < 3; ++i) u += G[xi][i] * (double)cv.w[((size_t)(d * 3 + i) * 128 + ci) * 128 + co];
                        U[(((size_t)d * 4 + xi) * 128 + ci) * 128 + co] = u;
                        mx = fmax(mx, fabs(u));
                    }
        int e = mx > 0.0 ? (int)floor(log2(16384.0 / mx)) : 0;
        e = e > 24 ? 24 : (e < -24 ? -24 : e);
        const double scale = ldexp(1.0, e);
        for (int d = 0; d < 3; ++d)
            for (int kk = 0; kk < 4; ++kk)
                for (int wv = 0; wv < 8; ++wv)
                    for (int xi = 0; xi < 4; ++xi)
                        for (int l = 0; l < 64; ++l)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int ci = 32 * kk + 8 * (l >> 4) + jj, co = 16 * wv + (l & 15);
                                const float v = (float)(U[(((size_t)d * 4 + xi) * 128 + ci) * 128 + co] * scale);
                                uint16_t hi, lo;
                                wsplit(v, hi, lo);
                                const size_t f0 = (size_t)li * layer_halfs + ((((size_t)(d * 4 + kk) * 8 + wv) * 4 + xi) * 2) * frag;
                                w[f0 + (size_t)l * 8 + jj] = hi;
                                w[f0 + frag + (size_t)l * 8 + jj] = lo;
                            }
        inv[li + 1] = (float)(1.0 / scale);
        for (int i = 0; i < 128; ++i) bias[(size_t)(li + 1) * 128 + i] = cv.bias[i];
    }
    std::vector<float> pt((size_t)128 * 65), vt((size_t)64 * 256);
    for (int o = 0; o < 65; ++o)
        for (int i = 0; i < 128; ++i) pt[(size_t)i * 65 + o] = hn.pfc_w[(size_t)o * 128 + i];
    for (int o = 0; o < 256; ++o)
        for (int i = 0; i < 64; ++i) vt[(size_t)i * 256 + o] = hn.vfc1_w[(size_t)o * 64 + i];
    OTH_HIP(hipMalloc(&ww->d_pfc_wt, pt.size() * 4));
    OTH_HIP(hipMalloc(&ww->d_vfc1_wt, vt.size() * 4));
    OTH_HIP(hipMemcpy(ww->d_pfc_wt, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_vfc1_wt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMalloc(&ww->d_w, w.size() * 2));
    OTH_HIP(hipMalloc(&ww->d_stem, stem.size() * 2));
    OTH_HIP(hipMalloc(&ww->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&ww->d_inv, inv.size() * 4));
    OTH_HIP(hipMemcpy(ww->d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_stem, stem.data(), stem.size() * 2, hipMemcpyHostToDevice));
    if (int rc = register_scaled_bias(net, ww->d_bias, std::move(bias))) return rc;
    OTH_HIP(hipMemcpy(ww->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    return OTH_OK;
}

// Launches that cannot fill the chip (every workgroup has a CU to itself either way) run one position per workgroup; both
// builds sum every accumulator in the same order, so a position's outputs do not depend on this.  OTH_WINO_TP=1|2 (read
// per call) forces a build: the switch the variants test uses to compare the two on the same batch.  The ONE place the
// decision is made: oth_net_kernel_info reports what this returns.
int wino_positions_per_workgroup(int64_t n) {
    const char* tpe = getenv("OTH_WINO_TP");
    return tpe ? (atoi(tpe) == 1 ? 1 : 2) : (n <= 256 ? 1 : 2);
}

int wino_forward(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
                 const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    OTH_CHECK(net->wino, "Winograd weights not packed");
    WinoArgs a;
    a.w = net->wino->d_w;
    a.stem = net->wino->d_stem;
    a.bias = net->wino->d_bias;
    a.inv = net->wino->d_inv;
    a.n_res_layers = 2 * net->wino->blocks;
    a.heads = net->heads;
    a.pfc_wt = net->wino->d_pfc_wt;
    a.vfc1_wt = net->wino->d_vfc1_wt;
    a.sat = net->d_sat;
    a.act_scale = net->act_scale;
    a.dbg = nullptr;
#ifdef OTH_STAMPS
    const unsigned dbg_grid = (unsigned)n;
    OTH_HIP(hipMalloc(&a.dbg, (size_t)dbg_grid * 8 * 8 * sizeof(unsigned long long)));
    OTH_HIP(hipMemset(a.dbg, 0, (size_t)dbg_grid * 8 * 8 * sizeof(unsigned long long)));
#endif
    static bool attr_set_dev[64] = {};
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_w<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kWLds));
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_w<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kWLds));
        attr_set = true;
    }
    const int tp = wino_positions_per_workgroup(n);
    const unsigned grid = (unsigned)((n + tp - 1) / tp);
    if (tp == 1) hipLaunchKernelGGL(k_trunk_w<1>, dim3(grid), dim3(512), kWLds, stream, a, sb, ob, lg, n, n_valid, logp, v);
    else hipLaunchKernelGGL(k_trunk_w<2>, dim3(grid), dim3(512), kWLds, stream, a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipGetLastError());
#ifdef OTH_STAMPS
    {
        OTH_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)dbg_grid * 8 * 8);
        OTH_HIP(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double sm[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long r0 = ~0ull, r1 = 0;
        for (size_t w = 0; w < (size_t)dbg_grid * 8; ++w) {
            for (int i = 0; i < 6; ++i) sm[i] += (double)h[w * 8 + i];
            if (!h[w * 8 + 4]) continue;
            if (h[w * 8 + 6] < r0) r0 = h[w * 8 + 6];
            if (h[w * 8 + 6] + h[w * 8 + 5] > r1) r1 = h[w * 8 + 6] + h[w * 8 + 5];
        }
        const double nw = (double)dbg_grid * 8;
        fprintf(stderr, "[wino stamps] in-kernel clock %.3f GHz; first start -> last trunk end %.3f ms\n", sm[4] / sm[5] * 0.1,
                (double)(r1 - r0) * 1e-5);
        double pro = 0;
        for (size_t w = 0; w < (size_t)dbg_grid * 8; ++w) pro += (double)h[w * 8 + 7];
#if OTH_STAMPS == 2
        fprintf(stderr, "[wino stamps] per-wave cycles: weight prefetch %.0f | barrier waits %.0f | epilogues %.0f | convolutions %.0f | layers total %.0f | prologue+stem %.0f\n",
                sm[0] / nw * 2, sm[1] / nw * 2, sm[2] / nw * 2, sm[3] / nw * 2, sm[4] / nw * 2, pro / nw * 2);
#else
        fprintf(stderr, "[wino stamps] per-wave cycles (x2: averaged over twice the workgroups): planes->LDS %.0f | heads stage 1 (1x1 convs) %.0f | FCs %.0f | heads all %.0f | layers total %.0f | prologue+stem %.0f\n",
                sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw, pro / nw);
#endif
        (void)hipFree(a.dbg);
    }
#endif
    return OTH_OK;
}

}  // namespace oth
