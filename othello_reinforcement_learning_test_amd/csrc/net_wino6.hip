// net_wino6.hip -- OthelloResNet forward for 64 filters on the 6x6 board (BASELINE configs[4]) with the residual 3x3
// convolutions as a 1-D Winograd F(2,3) along x, the same transform and fp16-split arithmetic as net_wino.hip (see its
// header for the algebra): 1.5x fewer MFMAs than the direct form, and no padded cells (k_trunk_h3 multiplies 80 columns for
// 72 cells).
//
// Reference: /root/reference/src/model/net.py:182-205 (eval mode; BatchNorm folded at load time).
//
// One 256-thread workgroup (4 waves) per CU carries EIGHT positions through the whole network.  Wave w owns output
// channels [16w, 16w+16).  A 6-cell row has three Winograd tiles; the MFMA column (lane & 15) is a (position, row) pair
// -- 8 x 6 = 48 pairs = three lane groups -- and the tile column j = 0..2 selects the ACCUMULATOR: N-tile nt = 3*lg + j,
// nine N-tiles x four transformed taps = 36 accumulators per wave.  The x neighbours in(2j-1), in(2j+2) of the input
// transform are therefore other registers of the SAME lane: no DPP, no edge masks (j = 0 and j = 2 are compile-time), and
// all nine N-tiles are full (48 = 3 x 16).  The transformed operand V lives in LDS as [tile = (position, row, j)][xi][hi 64
// x f16 | lo 64 x f16] = 144 KB for eight positions; the 16-byte slot of a lane's k-group is XOR-swizzled with twice the low
// bits of the (position, row) index, so that the lanes of every ds_read_b128 service group cover all 64 banks for all
// three row taps (see "V addressing" below).  Every weight is loaded once per EIGHT positions (k_trunk_h3: once per two).
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
using f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int k6F = 64, k6BS = 6, k6Cells = 36, k6NP = 37, k6TP = 8;
constexpr int k6NT = 9;                         // N-tiles of a wave: 3 lane groups x 3 tile columns
constexpr int k6TileBytes = 1024;               // one tile: 4 xi x (128 B hi + 128 B lo)
constexpr int k6VBytes = k6TP * 18 * k6TileBytes;   // 147456
constexpr int k6ZeroOff = k6VBytes;             // 4 KB of zeros: the source of out-of-board rows (any j, xi, k-step)
constexpr int k6Lds = k6VBytes + 4096;          // 151552
constexpr float k6ActScale = 16.0f;             // activations and residual are carried x 2^4
constexpr float k6Clamp = 30000.0f;             // |V| <= 2 x activation must stay in the f16 range: activations <= 1875
constexpr int k6Groups = 6;                     // (row tap d, k-step kk of 32 input channels): g = 2*d + kk
constexpr int k6GroupU4 = 4 * 8 * 64;           // uint4 per group: 4 waves x 8 fragments x 64 lanes

struct Wino6Weights {
    int blocks = 0;
    uint4* d_w = nullptr;     // [layer][group 6][wave 4][xi 4][hi, lo][64 lanes] x 16 B   (A fragments of U)
    uint4* d_stem = nullptr;  // [wave 4][hi, lo][64 lanes] x 16 B: direct 3x3 stem as one k-step of 32 (27 used)
    float* d_bias = nullptr;  // [1 + 2*blocks][64], x k6ActScale
    float* d_inv = nullptr;   // [1 + 2*blocks] 1 / weight scale
    float* d_pfc_wt = nullptr;   // [72][37]  policy FC transposed (net_heads_wave.h)
    float* d_vfc1_wt = nullptr;  // [36][256] value FC1 transposed
};

struct Wino6Args {
    const uint4* w;
    const uint4* stem;
    const float* bias;
    const float* inv;
    int n_res_layers;
    HeadParams heads;
    const float* pfc_wt;
    const float* vfc1_wt;
    int* sat;
    unsigned long long* dbg;   // diagnostic build (-DOTH_STAMPS) only: per-wave phase cycle sums
};

#ifdef OTH_STAMPS
__device__ __forceinline__ unsigned long long w6_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long w6_realclk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define OTH_W6STAMP(i) { const unsigned long long t1_ = w6_clk(); ph_[i] += t1_ - t0_; t0_ = t1_; }
#else
#define OTH_W6STAMP(i)
#endif

// The builtin, not the in-place inline asm of net_wino.hip: 36 accumulators + 18 residual registers + the weight ring need
// more than 256 architectural VGPRs, and with asm MFMAs hipcc parked WEIGHT fragments in AGPRs and copied them back right
// in front of their MFMA (no wait states: tools/check_mfma_hazards.py flagged every one).  With the builtin the allocator
// may keep the accumulators themselves in AGPRs (MFMA reads and writes them there) and knows the hazards.
// In-kernel stamps (-DOTH_STAMPS) of this build, per wave and layer: convolution 18.8 k cycles for 648 MFMAs (29 cycles per
// MFMA; 17.8 k with the conflict-free key below), epilogue 6.6 k.  Timing ablations with fewer N-tiles and VGPR accumulators: 21.6 cycles per MFMA -- with one wave
// per SIMD the two ds_read_b128 of a step cost their ~8 issue cycles each on top of its three MFMAs (48 + 16), and AGPR
// accumulators add ~7 cycles per MFMA.  Not cured by a longer LDS look-ahead (3 / 4 / 6 steps), by interleaving the MFMAs
// of two steps, or by asm MFMAs with the residual pinned into AGPRs (the allocator then migrates accumulators right after
// their MFMA: 64 hazards), nor by eight waves (k_trunk_w6b below: the SIMD's 648 MFMAs still take ~19 k cycles).  Open.
__device__ __forceinline__ f32x4 w6mfma(half8 a, half8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 w6mfma0(half8 a, half8 b) {   // first product of a chain: C = 0
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
__device__ __forceinline__ void w6barrier() {   // LDS-only barrier: global weight prefetches stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define OTH_W6SB __builtin_amdgcn_sched_barrier(0)

// All 512 registers per lane: nothing else is resident on the CU beside this workgroup -- in the two-lane engine the other
// lane's tree kernel (70 VGPRs) then runs after the trunk instead of beside it (trunk share of the step 0.98 -> 0.92).  Capped
// at 440 / 400 registers (-DOTH_W6REGS=220 / 200) the tree kernel is back beside it, but the spills cost more: 17.6 k / 17.0 k
// games/s on configs[4] against 23.1 k uncapped (k_trunk_h3: 21.9 k on the same box).
#ifndef OTH_W6REGS
#define OTH_W6REGS 256   // x 2 = total registers per lane (VGPR + AGPR)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(OTH_W6REGS))) void k_trunk_w6(Wino6Args a, const uint64_t* __restrict__ sb,
                                                  const uint64_t* __restrict__ ob, const uint64_t* __restrict__ lgl,
                                                  int64_t n, const int32_t* __restrict__ n_valid, float* __restrict__ logp,
                                                  float* __restrict__ vout) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int64_t pos0 = (int64_t)blockIdx.x * k6TP;
    if (pos0 >= nv) return;
#ifdef OTH_STAMPS
    unsigned long long ph_[5] = {0, 0, 0, 0, 0}, t0_ = w6_clk();   // prologue + stem | barrier waits | epilogues | convolutions | heads
    const unsigned long long tstart_ = t0_, rstart_ = w6_realclk();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, c = lane & 15;

    // ---- stem input: im2col of the three bit planes, [288 cells][32 k] f16 (64 B per cell) at LDS 0 (V is not live yet).
    //      The rows of odd-x output cells are NEGATED: the stem's result then has the form of a Winograd-domain accumulator
    //      set (M0 = y0, M1 = M2 = 0, M3 = -y1) and goes through the same epilogue as every other layer.
    for (int ci = tid; ci < k6TP * k6Cells; ci += 256) {
        const int p = ci / k6Cells, cell = ci % k6Cells, y = cell / k6BS, x = cell % k6BS;
        const bool live = pos0 + p < nv;
        const uint64_t b0 = live ? sb[pos0 + p] : 0, b1 = live ? ob[pos0 + p] : 0, b2 = live ? lgl[pos0 + p] : 0;
        const _Float16 one = (x & 1) ? (_Float16)(-k6ActScale) : (_Float16)k6ActScale;
        _Float16 vals[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) vals[i] = (_Float16)0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < k6BS && xx >= 0 && xx < k6BS;
            const int s = ok ? yy * k6BS + xx : 0;
            vals[tap * 3 + 0] = (ok && ((b0 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 1] = (ok && ((b1 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 2] = (ok && ((b2 >> s) & 1ULL)) ? one : (_Float16)0.0f;
        }
        half8* dst = (half8*)(lds + ci * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = vals[q * 8 + i];
            dst[q] = t;
        }
    }
    ((uint4*)(lds + k6ZeroOff))[tid] = make_uint4(0, 0, 0, 0);   // 4 KB of zeros
    __syncthreads();

    f32x4 acc[4][k6NT];   // [xi][N-tile]
    f32x4 res[k6NT][2];   // [N-tile][x parity]: the residual in the spatial domain, fp32, x 2^4
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nt = 0; nt < k6NT; ++nt) {
            acc[xi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            OTH_PIN_ACC(acc[xi][nt]);
        }
    OTH_PIN_ACC_END();
#pragma unroll
    for (int nt = 0; nt < k6NT; ++nt) res[nt][0] = res[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f};   // the stem "adds" to this

    // lane constants: the (position, row) pair of this lane's MFMA column in each of the three lane groups
    int pr_l[3], p_l[3], row_l[3];
#pragma unroll
    for (int lg = 0; lg < 3; ++lg) {
        pr_l[lg] = lg * 16 + c;
        p_l[lg] = pr_l[lg] / k6BS;
        row_l[lg] = pr_l[lg] % k6BS;
    }

    {   // ---- stem conv (net.py:195), direct: acc[0] = y0 (even-x cells) and acc[3] = -y1 (odd-x cells, negated rows)
        const uint4* wp = a.stem + (size_t)wave * 2 * 64 + lane;
        const half8 wh = __builtin_bit_cast(half8, wp[0]), wlo = __builtin_bit_cast(half8, wp[64]);
#pragma unroll
        for (int lg = 0; lg < 3; ++lg)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cell = p_l[lg] * k6Cells + row_l[lg] * k6BS + 2 * j + e;
                    const half8 xh = *(const half8*)(lds + cell * 64 + g4 * 16);
                    acc[3 * e][lg * 3 + j] = w6mfma(wlo, xh, acc[3 * e][lg * 3 + j]);
                    acc[3 * e][lg * 3 + j] = w6mfma(wh, xh, acc[3 * e][lg * 3 + j]);
                }
    }

    // V addressing.  Tile T = (p*6 + row)*3 + j at T*1024; inside a tile xi*256 + 16 * (slot ^ key) with slot = half*8 + chunk
    // (chunk = 4 kk + g4 for a reader) and key = 2 * (pr & 7) of the tile's (position, row) index pr.  ds_read_b128 is
    // served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), which mix two
    // k-groups; this key keeps the 16 slots of every group distinct for all three row taps (pr - 1, pr, pr + 1) -- found
    // by exhaustive search over that model.  (The first key, (pr & 7) with the halves swapped by bit 3, was conflict-free
    // only for the middle tap: 93 B/clk of LDS reads, and the convolution no faster on eight waves than on four.)  For a
    // fixed lane the k-step kk = 1 is the address XOR 64 and the lo half the address XOR 128; j and xi are immediate offsets.
    const int ch0 = wave * 16 + 4 * g4;                       // + r: this lane's four output channels
    const uint32_t wchunk = (uint32_t)(2 * wave + (g4 >> 1)); // 16-byte chunk of those channels (0..7)
    uint32_t wr_off[3];    // store address of (lane group, j = 0, xi = 0), hi half
    uint32_t rd_base[3][3];  // read address of (lane group, row tap), k-step 0, hi half, j = 0, xi = 0 -- or the zero block
#pragma unroll
    for (int lg = 0; lg < 3; ++lg) {
        const uint32_t pr = (uint32_t)pr_l[lg];
        wr_off[lg] = pr * 3u * k6TileBytes + ((wchunk << 4) ^ ((pr & 7u) << 5)) + 8u * (uint32_t)(g4 & 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int rs = row_l[lg] + d - 1;
            const uint32_t ps = (uint32_t)(pr_l[lg] + d - 1);   // (position, source row): same position when rs is on the board
            const uint32_t low = (((uint32_t)g4) << 4) ^ ((ps & 7u) << 5);
            rd_base[lg][d] = (rs >= 0 && rs < k6BS) ? ps * 3u * k6TileBytes + low : (uint32_t)k6ZeroOff + low;
        }
    }

    const int n_layers = 1 + a.n_res_layers;
    uint32_t sat_bits = 0;
    OTH_W6STAMP(0)
    uint4 wq[2][8];   // weight ring: [group parity][xi hi, xi lo]; a group = (row tap, k-step): 8 fragments
    float4 b4 = *(const float4*)(a.bias + ch0), b4n = b4;   // bias (x 2^4) and 1 / weight scale of the layer in the epilogue
    float inv = a.inv[0], invn = inv;
    for (int layer = 0; layer < n_layers; ++layer) {
        const bool last = layer == n_layers - 1;
        if (layer > 0) {   // loaded during the previous convolution
            b4 = b4n;
            inv = invn;
        }
        const f32x2 inv2 = {inv, inv};
        // A fragments of conv `layer+1`: group g at wl + g * k6GroupU4
        const uint4* wl = a.w + (size_t)layer * (k6Groups * k6GroupU4) + (size_t)wave * (8 * 64) + lane;
        // ---------------- epilogue of conv `layer`: output transform, scale, bias, skip, ReLU; then the next layer's
        //                  input transform and the hi/lo re-split into V.  Two variants (net_wino.hip): SKIP for the stem
        //                  (zero residual) and the second convolution of a block, plain for the first.
        auto epilogue = [&](auto SKIP) {
            constexpr bool add_res = decltype(SKIP)::value;
#pragma unroll
            for (int lg = 0; lg < 3; ++lg) {
                f32x2 v0[3][2], v1[3][2];   // [tile column j][channel pair]: outputs x = 2j, 2j+1
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int nt = lg * 3 + j;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 a0 = whalf(acc[0][nt], h), a1 = whalf(acc[1][nt], h), a2 = whalf(acc[2][nt], h),
                                    a3 = whalf(acc[3][nt], h);
                        const f32x2 bb = h == 0 ? f32x2{b4.x, b4.y} : f32x2{b4.z, b4.w};
                        const f32x2 t0 = pk_fma(pk_add(pk_add(a0, a1), a2), inv2, bb);
                        const f32x2 t1 = pk_fma(pk_sub(pk_sub(a1, a2), a3), inv2, bb);
                        if (add_res) {
                            f32x2 r0 = whalf(res[nt][0], h), r1 = whalf(res[nt][1], h);
                            pk_add_relu_inplace(r0, t0, k6Clamp);
                            pk_add_relu_inplace(r1, t1, k6Clamp);
                            wsethalf(res[nt][0], h, r0);
                            wsethalf(res[nt][1], h, r1);
                            v0[j][h] = r0;
                            v1[j][h] = r1;
                        } else {
                            v0[j][h] = f32x2{__builtin_amdgcn_fmed3f(t0.x, 0.f, k6Clamp), __builtin_amdgcn_fmed3f(t0.y, 0.f, k6Clamp)};
                            v1[j][h] = f32x2{__builtin_amdgcn_fmed3f(t1.x, 0.f, k6Clamp), __builtin_amdgcn_fmed3f(t1.y, 0.f, k6Clamp)};
                        }
                        sat_bits = max(sat_bits, max(max(__float_as_uint(v0[j][h].x), __float_as_uint(v0[j][h].y)),
                                                     max(__float_as_uint(v1[j][h].x), __float_as_uint(v1[j][h].y))));
                    }
                }
                if (!last) {
                    const f32x2 zero2 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        uint32_t wo = wr_off[lg] + j * k6TileBytes;
                        asm volatile("" : "+v"(wo));   // one base register + immediate offsets
                        const uint32_t wol = wo ^ 128u;  // the lo half of the same chunk
                        f32x2 V[4][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            // d0 = in(2j-1) = v1[j-1], d1 = v0[j], d2 = v1[j], d3 = in(2j+2) = v0[j+1]; board edges are zeros
                            const f32x2 d0 = j > 0 ? v1[j > 0 ? j - 1 : 0][h] : zero2, d3 = j < 2 ? v0[j < 2 ? j + 1 : 2][h] : zero2;
                            V[0][h] = pk_sub(d0, v1[j][h]);
                            V[1][h] = pk_add(v0[j][h], v1[j][h]);
                            V[2][h] = pk_sub(v1[j][h], v0[j][h]);
                            V[3][h] = pk_sub(v0[j][h], d3);
                        }
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi) {
                            uint2 hi, lo;
                            hi.x = wpack(V[xi][0].x, V[xi][0].y);
                            hi.y = wpack(V[xi][1].x, V[xi][1].y);
                            lo.x = wresid(hi.x, V[xi][0].x, V[xi][0].y);
                            lo.y = wresid(hi.y, V[xi][1].x, V[xi][1].y);
                            *(uint2*)(lds + wo + xi * 256) = hi;
                            *(uint2*)(lds + wol + xi * 256) = lo;
                        }
                    }
                }
                OTH_W6SB;   // one lane group at a time (without it: no change)
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (layer == 0 && !last) {   // the first weight group of the first convolution; the later ones load theirs in
                                     // the previous convolution's last group
#pragma unroll
            for (int f = 0; f < 8; ++f) wq[0][f] = wl[(size_t)f * 64];
        }
        OTH_W6STAMP(3)
        w6barrier();   // every wave has finished reading V (or the stem's im2col)
        OTH_W6STAMP(1)
        if (layer & 1) epilogue(F_{});
        else epilogue(T_{});
        OTH_W6STAMP(2)
        if (last) break;
        w6barrier();
        OTH_W6STAMP(1)

        // ---------------- conv `layer+1` in the Winograd domain: 6 groups (row tap d, k-step kk) x 36 steps (N-tile, xi) x
        // 3 split products.  One straight-line software pipeline per row tap: the two LDS reads of step q+2 and, spread
        // over a group, the eight weight loads of the next group sit between the MFMAs.  step q = (g*9 + nt)*4 + xi.
        constexpr int GS = k6NT * 4;            // steps of a group
        constexpr int QT = k6Groups * GS;       // steps of a layer
        auto src_of = [&](int q) -> uint32_t {
            const int xi = q & 3, nt = (q >> 2) % k6NT, grp = q / GS, kk = grp & 1, d = grp >> 1;
            return (rd_base[nt / 3][d] ^ (uint32_t)(kk << 6)) + (uint32_t)((nt % 3) * k6TileBytes + xi * 256);
        };
#ifndef OTH_W6PD
#define OTH_W6PD 2
#endif
        constexpr int PD = OTH_W6PD;   // LDS operand pairs in flight ahead of the MFMAs (steps)
        half8 xh[PD + 1], xl[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const uint32_t s = src_of(q);
            xh[q] = *(const half8*)(lds + s);
            xl[q] = *(const half8*)(lds + (s ^ 128u));
        }
        auto conv_d = [&](auto DC) {
            constexpr int D = decltype(DC)::value;
#pragma unroll
            for (int ql = 0; ql < 2 * GS; ++ql) {
                const int q = D * 2 * GS + ql;
                const int xi = q & 3, nt = (q >> 2) % k6NT, grp = q / GS, sl = q % (PD + 1), psl = (q + PD) % (PD + 1);
                const int step = q % GS;
                const half8 wh = __builtin_bit_cast(half8, wq[grp & 1][2 * xi]);
                const half8 wlo = __builtin_bit_cast(half8, wq[grp & 1][2 * xi + 1]);
                OTH_W6SB;
                if (q < GS) acc[xi][nt] = w6mfma0(wh, xl[sl]);     // the layer's first group starts every accumulator
                else acc[xi][nt] = w6mfma(wh, xl[sl], acc[xi][nt]);
                OTH_W6SB;
                if (q + PD < QT) xh[psl] = *(const half8*)(lds + src_of(q + PD));
                OTH_W6SB;
                acc[xi][nt] = w6mfma(wh, xh[sl], acc[xi][nt]);
                OTH_W6SB;
                if (q + PD < QT) xl[psl] = *(const half8*)(lds + (src_of(q + PD) ^ 128u));
                if (grp == k6Groups - 2 && step == GS / 2) {
                    b4n = *(const float4*)(a.bias + (layer + 1) * k6F + ch0);
                    invn = a.inv[layer + 1];
                }
                // next group's fragments, one per step from the group's first step on (the last group loads group 0 of
                // the NEXT convolution: the layers are contiguous and one zero group pads the end of the array)
                if (step < 8) wq[(grp + 1) & 1][step] = wl[(size_t)(grp + 1) * k6GroupU4 + (size_t)step * 64];
                OTH_W6SB;
                acc[xi][nt] = w6mfma(wlo, xh[sl], acc[xi][nt]);
                OTH_W6SB;
            }
        };
        conv_d(std::integral_constant<int, 0>{});
        conv_d(std::integral_constant<int, 1>{});
        conv_d(std::integral_constant<int, 2>{});
    }

    // ---------------- heads (fp32 VALU): final activations (in `res`, x 2^4) -> LDS planes [channel][8 x 36 cells] f32
    //                  (aliasing V: every read of it is done), then each wave runs the shared one-wave head code on two
    //                  of the eight positions
    if (sat_bits >= __float_as_uint(k6Clamp)) atomicOr(a.sat, 1);
    __syncthreads();
    constexpr int NCO = k6TP * k6Cells;   // 288
    float* planes = (float*)lds;
#pragma unroll
    for (int lg = 0; lg < 3; ++lg)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ci = p_l[lg] * k6Cells + row_l[lg] * k6BS + 2 * j + e;
                const f32x4 v = res[lg * 3 + j][e];
#pragma unroll
                for (int r = 0; r < 4; ++r) planes[(ch0 + r) * NCO + ci] = v[r] * (1.0f / k6ActScale);
            }
    __syncthreads();
    {
        float* scratch = (float*)(lds + (size_t)k6F * NCO * 4) + wave * 2 * 192;
        const int cc = lane < k6Cells ? lane : 0;
        const float* srcs[2];
        float* lps[2];
        float* vs[2];
        bool live[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = 2 * wave + k;
            srcs[k] = planes + p * k6Cells + cc;
            lps[k] = logp + (pos0 + p) * k6NP;
            vs[k] = vout + pos0 + p;
            live[k] = pos0 + p < nv;
        }
        heads_wave_n<k6F, k6BS, 2>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, NCO, scratch, lane, lps, vs, live);
    }
#ifdef OTH_STAMPS
    OTH_W6STAMP(4)
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ph_[i];
        o[5] = w6_clk() - tstart_;
        o[6] = w6_realclk() - rstart_;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_trunk_w6b (experiment, OTH_WINO6=2; correct -- the parity and reproducibility tests pass with it -- but not faster: 0.255
// vs 0.244 ms per 4096 positions; stamps: convolutions 13.4 k + barriers / exchange 6.1-6.7 k + epilogue 4.3-5.8 k cycles per
// layer): the same network and arithmetic on EIGHT waves (two per SIMD).  Wave (cb, xp): output channels [16 cb,
// 16 cb + 16), transformed taps xi = 2 xp and 2 xp + 1 -- 18 accumulators in VGPRs (in-place asm MFMAs again), half the
// weight fragments and half the operand reads of a four-wave wave, and a partner on its SIMD to fill the issue slots of
// its ds_read_b128s.  The output transform needs all four taps, so after a convolution every wave publishes its
// accumulators in the (then dead) V buffer, and takes from its partner's the halves it needs: wave xp finishes channel
// pair h = xp of the lane's four channels for all nine N-tiles (bias, skip, ReLU, input transform, re-split), i.e. the
// epilogue's work is split evenly too.  Four barriers per layer instead of two.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 w6bmfma(half8 a, half8 b, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x4 w6bmfma0(half8 a, half8 b) {
    f32x4 c;
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
    return c;
}

__global__ __launch_bounds__(512, 2) void k_trunk_w6b(Wino6Args a, const uint64_t* __restrict__ sb,
                                                      const uint64_t* __restrict__ ob, const uint64_t* __restrict__ lgl,
                                                      int64_t n, const int32_t* __restrict__ n_valid,
                                                      float* __restrict__ logp, float* __restrict__ vout) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int64_t pos0 = (int64_t)blockIdx.x * k6TP;
    if (pos0 >= nv) return;
#ifdef OTH_STAMPS
    unsigned long long ph_[5] = {0, 0, 0, 0, 0}, t0_ = w6_clk();   // prologue + stem | barriers + exchange | epilogues | convolutions | heads
    const unsigned long long tstart_ = t0_, rstart_ = w6_realclk();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 3, xp = wave >> 2;   // channel block, tap pair
    const int g4 = lane >> 4, c = lane & 15;

    // ---- stem input: im2col as in k_trunk_w6 (rows of odd-x output cells negated)
    for (int ci = tid; ci < k6TP * k6Cells; ci += 512) {
        const int p = ci / k6Cells, cell = ci % k6Cells, y = cell / k6BS, x = cell % k6BS;
        const bool live = pos0 + p < nv;
        const uint64_t b0 = live ? sb[pos0 + p] : 0, b1 = live ? ob[pos0 + p] : 0, b2 = live ? lgl[pos0 + p] : 0;
        const _Float16 one = (x & 1) ? (_Float16)(-k6ActScale) : (_Float16)k6ActScale;
        _Float16 vals[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) vals[i] = (_Float16)0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < k6BS && xx >= 0 && xx < k6BS;
            const int s = ok ? yy * k6BS + xx : 0;
            vals[tap * 3 + 0] = (ok && ((b0 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 1] = (ok && ((b1 >> s) & 1ULL)) ? one : (_Float16)0.0f;
            vals[tap * 3 + 2] = (ok && ((b2 >> s) & 1ULL)) ? one : (_Float16)0.0f;
        }
        half8* dst = (half8*)(lds + ci * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 t;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = vals[q * 8 + i];
            dst[q] = t;
        }
    }
    if (tid < 256) ((uint4*)(lds + k6ZeroOff))[tid] = make_uint4(0, 0, 0, 0);   // 4 KB of zeros
    __syncthreads();

    f32x4 acc[2][k6NT];   // [local tap][N-tile]
    f32x2 res[k6NT][2];   // [N-tile][x parity]: this wave's channel pair of the residual, fp32, x 2^4
#pragma unroll
    for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int nt = 0; nt < k6NT; ++nt) {
            acc[xl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            OTH_PIN_ACC(acc[xl][nt]);
        }
    OTH_PIN_ACC_END();
#pragma unroll
    for (int nt = 0; nt < k6NT; ++nt) res[nt][0] = res[nt][1] = f32x2{0.f, 0.f};

    int pr_l[3], p_l[3], row_l[3];
#pragma unroll
    for (int lg = 0; lg < 3; ++lg) {
        pr_l[lg] = lg * 16 + c;
        p_l[lg] = pr_l[lg] / k6BS;
        row_l[lg] = pr_l[lg] % k6BS;
    }

    {   // ---- stem conv: wave xp = 0 computes M0 = y0 (even-x cells) into its local tap 0, wave xp = 1 computes M3 = -y1
        //      (odd-x cells, negated rows) into its local tap 1; the other local tap stays zero (M1 = M2 = 0)
        const uint4* wp = a.stem + (size_t)cb * 2 * 64 + lane;
        const half8 wh = __builtin_bit_cast(half8, wp[0]), wlo = __builtin_bit_cast(half8, wp[64]);
#pragma unroll
        for (int lg = 0; lg < 3; ++lg)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int cell = p_l[lg] * k6Cells + row_l[lg] * k6BS + 2 * j + xp;
                const half8 xh = *(const half8*)(lds + cell * 64 + g4 * 16);
                if (xp == 0) {
                    acc[0][lg * 3 + j] = w6bmfma(wlo, xh, acc[0][lg * 3 + j]);
                    acc[0][lg * 3 + j] = w6bmfma(wh, xh, acc[0][lg * 3 + j]);
                } else {
                    acc[1][lg * 3 + j] = w6bmfma(wlo, xh, acc[1][lg * 3 + j]);
                    acc[1][lg * 3 + j] = w6bmfma(wh, xh, acc[1][lg * 3 + j]);
                }
            }
    }

    // V addressing as in k_trunk_w6; this wave's taps are xi = 2 xp + xl: 512 xp is folded into the bases
    const int ch0 = cb * 16 + 4 * g4 + 2 * xp;                // + r (0, 1): this wave's two output channels of the lane's four
    const uint32_t wchunk = (uint32_t)(2 * cb + (g4 >> 1));
    uint32_t wr_off[3];      // store address of (lane group, j = 0, xi = 0), hi half, this wave's channel pair
    uint32_t rd_base[3][3];  // read address of (lane group, row tap), k-step 0, hi half, j = 0, local tap 0
#pragma unroll
    for (int lg = 0; lg < 3; ++lg) {
        const uint32_t pr = (uint32_t)pr_l[lg];
        wr_off[lg] = pr * 3u * k6TileBytes + ((wchunk << 4) ^ ((pr & 7u) << 5)) + 8u * (uint32_t)(g4 & 1) +
                     4u * (uint32_t)xp;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int rs = row_l[lg] + d - 1;
            const uint32_t ps = (uint32_t)(pr_l[lg] + d - 1);
            const uint32_t low = (((uint32_t)g4) << 4) ^ ((ps & 7u) << 5);
            rd_base[lg][d] = ((rs >= 0 && rs < k6BS) ? ps * 3u * k6TileBytes + low : (uint32_t)k6ZeroOff + low) + 512u * (uint32_t)xp;
        }
    }
    // exchange buffer (aliases V while it is dead): accumulator (wave W, local tap xl, N-tile nt) at ((W*18 + xl*9 + nt)*64 + lane)*16
    const uint32_t x_mine = (uint32_t)((wave * 18) * 64 + lane) * 16u;
    const uint32_t x_part = (uint32_t)(((wave ^ 4) * 18) * 64 + lane) * 16u + 8u * (uint32_t)xp;   // the partner's, my channel pair

    const int n_layers = 1 + a.n_res_layers;
    uint32_t sat_bits = 0;
    OTH_W6STAMP(0)
    uint4 wq[2][4];   // weight ring: [group parity][local tap hi, lo]
    float2 b2 = *(const float2*)(a.bias + ch0), b2n = b2;
    float inv = a.inv[0], invn = inv;
    for (int layer = 0; layer < n_layers; ++layer) {
        const bool last = layer == n_layers - 1;
        if (layer > 0) {
            b2 = b2n;
            inv = invn;
        }
        const f32x2 inv2 = {inv, inv}, bb = {b2.x, b2.y};
        // A fragments of conv `layer+1`: group g at wl + g * k6GroupU4; this wave's are fragments 4 xp .. 4 xp + 3 of its block
        const uint4* wl = a.w + (size_t)layer * (k6Groups * k6GroupU4) + (size_t)cb * (8 * 64) + (size_t)(4 * xp) * 64 + lane;
        if (layer == 0 && !last) {
#pragma unroll
            for (int f = 0; f < 4; ++f) wq[0][f] = wl[(size_t)f * 64];
        }
        OTH_W6STAMP(3)
        w6barrier();   // B1: every wave has finished reading V (or the stem's im2col)
        // publish the accumulators
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int nt = 0; nt < k6NT; ++nt) *(f32x4*)(lds + x_mine + (uint32_t)((xl * 9 + nt) * 1024)) = acc[xl][nt];
        w6barrier();   // B2
        f32x2 pm[2][k6NT];   // the partner's taps, my channel pair
#pragma unroll
        for (int xl = 0; xl < 2; ++xl)
#pragma unroll
            for (int nt = 0; nt < k6NT; ++nt) pm[xl][nt] = *(const f32x2*)(lds + x_part + (uint32_t)((xl * 9 + nt) * 1024));
        w6barrier();   // B3: the exchange buffer is read; V may be written
        OTH_W6STAMP(1)
        auto epilogue = [&](auto SKIP) {
            constexpr bool add_res = decltype(SKIP)::value;
#pragma unroll
            for (int lg = 0; lg < 3; ++lg) {
                f32x2 v0[3], v1[3];   // [tile column j]: outputs x = 2j, 2j+1, this wave's channel pair
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int nt = lg * 3 + j;
                    const f32x2 o0 = whalf(acc[0][nt], xp), o1 = whalf(acc[1][nt], xp);   // my taps 2 xp, 2 xp + 1
                    const f32x2 m0 = xp == 0 ? o0 : pm[0][nt], m1 = xp == 0 ? o1 : pm[1][nt];
                    const f32x2 m2 = xp == 0 ? pm[0][nt] : o0, m3 = xp == 0 ? pm[1][nt] : o1;
                    const f32x2 t0 = pk_fma(pk_add(pk_add(m0, m1), m2), inv2, bb);
                    const f32x2 t1 = pk_fma(pk_sub(pk_sub(m1, m2), m3), inv2, bb);
                    if (add_res) {
                        pk_add_relu_inplace(res[nt][0], t0, k6Clamp);
                        pk_add_relu_inplace(res[nt][1], t1, k6Clamp);
                        v0[j] = res[nt][0];
                        v1[j] = res[nt][1];
                    } else {
                        v0[j] = f32x2{__builtin_amdgcn_fmed3f(t0.x, 0.f, k6Clamp), __builtin_amdgcn_fmed3f(t0.y, 0.f, k6Clamp)};
                        v1[j] = f32x2{__builtin_amdgcn_fmed3f(t1.x, 0.f, k6Clamp), __builtin_amdgcn_fmed3f(t1.y, 0.f, k6Clamp)};
                    }
                    sat_bits = max(sat_bits, max(max(__float_as_uint(v0[j].x), __float_as_uint(v0[j].y)),
                                                 max(__float_as_uint(v1[j].x), __float_as_uint(v1[j].y))));
                }
                if (!last) {
                    const f32x2 zero2 = {0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        uint32_t wo = wr_off[lg] + j * k6TileBytes;
                        asm volatile("" : "+v"(wo));
                        const uint32_t wol = wo ^ 128u;
                        const f32x2 d0 = j > 0 ? v1[j > 0 ? j - 1 : 0] : zero2, d3 = j < 2 ? v0[j < 2 ? j + 1 : 2] : zero2;
                        f32x2 V[4];
                        V[0] = pk_sub(d0, v1[j]);
                        V[1] = pk_add(v0[j], v1[j]);
                        V[2] = pk_sub(v1[j], v0[j]);
                        V[3] = pk_sub(v0[j], d3);
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi) {
                            const uint32_t hi = wpack(V[xi].x, V[xi].y);
                            const uint32_t lo = wresid(hi, V[xi].x, V[xi].y);
                            *(uint32_t*)(lds + wo + xi * 256) = hi;
                            *(uint32_t*)(lds + wol + xi * 256) = lo;
                        }
                    }
                }
                OTH_W6SB;
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (layer & 1) epilogue(F_{});
        else epilogue(T_{});
        OTH_W6STAMP(2)
        if (last) break;
        w6barrier();   // B4
        OTH_W6STAMP(1)

        // ---------------- conv `layer+1`: 6 groups x 18 steps (N-tile, local tap) x 3 split products
        constexpr int GS = k6NT * 2;
        constexpr int QT = k6Groups * GS;
        auto src_of = [&](int q) -> uint32_t {
            const int xl = q & 1, nt = (q >> 1) % k6NT, grp = q / GS, kk = grp & 1, d = grp >> 1;
            return (rd_base[nt / 3][d] ^ (uint32_t)(kk << 6)) + (uint32_t)((nt % 3) * k6TileBytes + xl * 256);
        };
        constexpr int PD = 2;
        half8 xh[PD + 1], xl_[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const uint32_t s0 = src_of(q);
            xh[q] = *(const half8*)(lds + s0);
            xl_[q] = *(const half8*)(lds + (s0 ^ 128u));
        }
        auto conv_d = [&](auto DC) {
            constexpr int D = decltype(DC)::value;
#pragma unroll
            for (int ql = 0; ql < 2 * GS; ++ql) {
                const int q = D * 2 * GS + ql;
                const int xl = q & 1, nt = (q >> 1) % k6NT, grp = q / GS, sl = q % (PD + 1), psl = (q + PD) % (PD + 1);
                const int step = q % GS;
#ifdef OTH_W6B_NTLO   // timing ablation only (wrong results): N-tiles [OTH_W6B_NTLO, OTH_W6B_NTHI) only
                if (nt < OTH_W6B_NTLO || nt >= OTH_W6B_NTHI) continue;
#endif
                const half8 wh = __builtin_bit_cast(half8, wq[grp & 1][2 * xl]);
                const half8 wlo = __builtin_bit_cast(half8, wq[grp & 1][2 * xl + 1]);
                OTH_W6SB;
                if (q < GS) acc[xl][nt] = w6bmfma0(wh, xl_[sl]);
                else acc[xl][nt] = w6bmfma(wh, xl_[sl], acc[xl][nt]);
                OTH_W6SB;
                if (q + PD < QT) xh[psl] = *(const half8*)(lds + src_of(q + PD));
                OTH_W6SB;
                acc[xl][nt] = w6bmfma(wh, xh[sl], acc[xl][nt]);
                OTH_W6SB;
                if (q + PD < QT) xl_[psl] = *(const half8*)(lds + (src_of(q + PD) ^ 128u));
                if (grp == k6Groups - 2 && step == GS / 2) {
                    b2n = *(const float2*)(a.bias + (layer + 1) * k6F + ch0);
                    invn = a.inv[layer + 1];
                }
                if (step < 4) wq[(grp + 1) & 1][step] = wl[(size_t)(grp + 1) * k6GroupU4 + (size_t)step * 64];
                OTH_W6SB;
                acc[xl][nt] = w6bmfma(wlo, xh[sl], acc[xl][nt]);
                OTH_W6SB;
            }
        };
        conv_d(std::integral_constant<int, 0>{});
        conv_d(std::integral_constant<int, 1>{});
        conv_d(std::integral_constant<int, 2>{});
    }

    // ---------------- heads: final activations (in `res`, this wave's channel pair) -> LDS planes, then ONE position per wave
    if (sat_bits >= __float_as_uint(k6Clamp)) atomicOr(a.sat, 1);
    __syncthreads();
    constexpr int NCO = k6TP * k6Cells;   // 288
    float* planes = (float*)lds;
#pragma unroll
    for (int lg = 0; lg < 3; ++lg)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ci = p_l[lg] * k6Cells + row_l[lg] * k6BS + 2 * j + e;
                const f32x2 v = res[lg * 3 + j][e];
                planes[(ch0 + 0) * NCO + ci] = v.x * (1.0f / k6ActScale);
                planes[(ch0 + 1) * NCO + ci] = v.y * (1.0f / k6ActScale);
            }
    __syncthreads();
    {
        float* scratch = (float*)(lds + (size_t)k6F * NCO * 4) + wave * 192;
        const int cc = lane < k6Cells ? lane : 0;
        const float* srcs[1] = {planes + wave * k6Cells + cc};
        float* lps[1] = {logp + (pos0 + wave) * k6NP};
        float* vs[1] = {vout + pos0 + wave};
        const bool live[1] = {pos0 + wave < nv};
        heads_wave_n<k6F, k6BS, 1>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, NCO, scratch, lane, lps, vs, live);
    }
#ifdef OTH_STAMPS
    OTH_W6STAMP(4)
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ph_[i];
        o[5] = w6_clk() - tstart_;
        o[6] = w6_realclk() - rstart_;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// host: transformed weights in A-fragment order
// ------------------------------------------------------------------------------------------------
static inline void w6split(float v, uint16_t& hi, uint16_t& lo) {
    const _Float16 hh = (_Float16)v;
    const _Float16 ll = (_Float16)(v - (float)hh);
    memcpy(&hi, &hh, 2);
    memcpy(&lo, &ll, 2);
}

void wino6_free_weights(oth_net* net) {
    if (!net->wino6) return;
    Wino6Weights* w = net->wino6;
    if (w->d_w) (void)hipFree(w->d_w);
    if (w->d_stem) (void)hipFree(w->d_stem);
    if (w->d_bias) (void)hipFree(w->d_bias);
    if (w->d_inv) (void)hipFree(w->d_inv);
    if (w->d_pfc_wt) (void)hipFree(w->d_pfc_wt);
    if (w->d_vfc1_wt) (void)hipFree(w->d_vfc1_wt);
    delete w;
    net->wino6 = nullptr;
}

int wino6_pack_weights(oth_net* net) {
    const HostNet& hn = net->host;
    OTH_CHECK(hn.filters == k6F && hn.board == k6BS, "the 6x6 Winograd trunk is built for 64 filters on 6x6");
    const int L = 2 * hn.blocks;
    Wino6Weights* ww = new Wino6Weights();
    ww->blocks = hn.blocks;
    net->wino6 = ww;
    const size_t frag = 64 * 8;                                     // halfs per fragment
    const size_t layer_halfs = (size_t)k6Groups * 4 * 8 * frag;     // 6 groups x 4 waves x (4 xi x hi/lo)
    std::vector<uint16_t> w((size_t)L * layer_halfs + 4 * 8 * frag), stem((size_t)4 * 2 * frag);   // + one zero group
    std::vector<float> bias((size_t)(L + 1) * k6F), inv(L + 1);
    {   // stem: direct, gemm k = tap*3 + plane (27 of 32), rows = 16 channels of a wave
        const FoldedConv& cv = hn.stem;
        float mx = 0.f;
        for (float x : cv.w) mx = fmaxf(mx, fabsf(x));
        int e = mx > 0.f ? (int)floorf(log2f(16384.0f / mx)) : 0;
        e = e > 24 ? 24 : (e < -24 ? -24 : e);
        const float scale = ldexpf(1.0f, e);
        for (int wv = 0; wv < 4; ++wv)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const int k = 8 * (l >> 4) + jj, co = 16 * wv + (l & 15);
                    const float v = k < 27 ? cv.w[(size_t)k * cv.cout + co] * scale : 0.f;   // [tap][cin=3][cout]
                    uint16_t hi, lo;
                    w6split(v, hi, lo);
                    stem[((size_t)wv * 2 + 0) * frag + (size_t)l * 8 + jj] = hi;
                    stem[((size_t)wv * 2 + 1) * frag + (size_t)l * 8 + jj] = lo;
                }
        inv[0] = 1.0f / scale;
        for (int i = 0; i < k6F; ++i) bias[i] = cv.bias[i] * k6ActScale;
    }
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    std::vector<double> U((size_t)3 * 4 * k6F * k6F);   // [dy][xi][ci][co]
    for (int li = 0; li < L; ++li) {
        const FoldedConv& cv = hn.res[li];
        double mx = 0.0;
        for (int d = 0; d < 3; ++d)
            for (int xi = 0; xi < 4; ++xi)
                for (int ci = 0; ci < k6F; ++ci)
                    for (int co = 0; co < k6F; ++co) {
                        double u = 0.0;
                        for (int i = 0; i < 3; ++i) u += G[xi][i] * (double)cv.w[((size_t)(d * 3 + i) * k6F + ci) * k6F + co];
                        U[(((size_t)d * 4 + xi) * k6F + ci) * k6F + co] = u;
                        mx = fmax(mx, fabs(u));
                    }
        int e = mx > 0.0 ? (int)floor(log2(16384.0 / mx)) : 0;
        e = e > 24 ? 24 : (e < -24 ? -24 : e);
        const double scale = ldexp(1.0, e);
        for (int d = 0; d < 3; ++d)
            for (int kk = 0; kk < 2; ++kk)
                for (int wv = 0; wv < 4; ++wv)
                    for (int xi = 0; xi < 4; ++xi)
                        for (int l = 0; l < 64; ++l)
                            for (int jj = 0; jj < 8; ++jj) {
                                const int ci = 32 * kk + 8 * (l >> 4) + jj, co = 16 * wv + (l & 15);
                                const float v = (float)(U[(((size_t)d * 4 + xi) * k6F + ci) * k6F + co] * scale);
                                uint16_t hi, lo;
                                w6split(v, hi, lo);
                                const size_t f0 = (size_t)li * layer_halfs + ((((size_t)(d * 2 + kk) * 4 + wv) * 4 + xi) * 2) * frag;
                                w[f0 + (size_t)l * 8 + jj] = hi;
                                w[f0 + frag + (size_t)l * 8 + jj] = lo;
                            }
        inv[li + 1] = (float)(1.0 / scale);
        for (int i = 0; i < k6F; ++i) bias[(size_t)(li + 1) * k6F + i] = cv.bias[i] * k6ActScale;
    }
    std::vector<float> pt((size_t)2 * k6Cells * k6NP), vt((size_t)k6Cells * 256);
    for (int o = 0; o < k6NP; ++o)
        for (int i = 0; i < 2 * k6Cells; ++i) pt[(size_t)i * k6NP + o] = hn.pfc_w[(size_t)o * 2 * k6Cells + i];
    for (int o = 0; o < 256; ++o)
        for (int i = 0; i < k6Cells; ++i) vt[(size_t)i * 256 + o] = hn.vfc1_w[(size_t)o * k6Cells + i];
    OTH_HIP(hipMalloc(&ww->d_pfc_wt, pt.size() * 4));
    OTH_HIP(hipMalloc(&ww->d_vfc1_wt, vt.size() * 4));
    OTH_HIP(hipMalloc(&ww->d_w, w.size() * 2));
    OTH_HIP(hipMalloc(&ww->d_stem, stem.size() * 2));
    OTH_HIP(hipMalloc(&ww->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&ww->d_inv, inv.size() * 4));
    OTH_HIP(hipMemcpy(ww->d_pfc_wt, pt.data(), pt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_vfc1_wt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_stem, stem.data(), stem.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(ww->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    return OTH_OK;
}

int wino6_forward(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
                  const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    OTH_CHECK(net->wino6, "6x6 Winograd weights not packed");
    Wino6Args a;
    memset(&a, 0, sizeof(a));
    a.w = net->wino6->d_w;
    a.stem = net->wino6->d_stem;
    a.bias = net->wino6->d_bias;
    a.inv = net->wino6->d_inv;
    a.n_res_layers = 2 * net->wino6->blocks;
    a.heads = net->heads;
    a.pfc_wt = net->wino6->d_pfc_wt;
    a.vfc1_wt = net->wino6->d_vfc1_wt;
    a.sat = net->d_sat;
    static bool attr_set_dev[64] = {};
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_w6, hipFuncAttributeMaxDynamicSharedMemorySize, k6Lds));
        attr_set = true;
    }
    const unsigned grid = (unsigned)((n + k6TP - 1) / k6TP);
    {
        const char* e = getenv("OTH_WINO6");   // read per call: the variants test toggles it
        if (e && atoi(e) == 2) {   // the eight-wave build (experiment switch)
            static bool attr_b[64] = {};
            if (!attr_b[net->device & 63]) {
                OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_w6b, hipFuncAttributeMaxDynamicSharedMemorySize, k6Lds));
                attr_b[net->device & 63] = true;
            }
#ifdef OTH_STAMPS
            OTH_HIP(hipMalloc(&a.dbg, (size_t)grid * 8 * 8 * sizeof(unsigned long long)));
            OTH_HIP(hipMemset(a.dbg, 0, (size_t)grid * 8 * 8 * sizeof(unsigned long long)));
#endif
            hipLaunchKernelGGL(k_trunk_w6b, dim3(grid), dim3(512), k6Lds, stream, a, sb, ob, lg, n, n_valid, logp, v);
            OTH_HIP(hipGetLastError());
#ifdef OTH_STAMPS
            OTH_HIP(hipStreamSynchronize(stream));
            {
                std::vector<unsigned long long> h((size_t)grid * 8 * 8);
                OTH_HIP(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
                double sm[7] = {0, 0, 0, 0, 0, 0, 0};
                size_t nw = 0;
                for (size_t w = 0; w < (size_t)grid * 8; ++w) {
                    if (!h[w * 8 + 5]) continue;
                    ++nw;
                    for (int i = 0; i < 7; ++i) sm[i] += (double)h[w * 8 + i];
                }
                fprintf(stderr, "[w6b stamps] per-wave cycles: prologue+stem %.0f | barriers + exchange %.0f | epilogues %.0f | convolutions %.0f | heads %.0f | total %.0f | clock %.3f GHz\n",
                        sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw, sm[5] / nw, sm[5] / sm[6] * 0.1);
                (void)hipFree(a.dbg);
            }
#endif
            return OTH_OK;
        }
    }
#ifdef OTH_STAMPS
    OTH_HIP(hipMalloc(&a.dbg, (size_t)grid * 4 * 8 * sizeof(unsigned long long)));
    OTH_HIP(hipMemset(a.dbg, 0, (size_t)grid * 4 * 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_trunk_w6, dim3(grid), dim3(256), k6Lds, stream, a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipStreamSynchronize(stream));
    {
        std::vector<unsigned long long> h((size_t)grid * 4 * 8);
        OTH_HIP(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double sm[7] = {0, 0, 0, 0, 0, 0, 0};
        size_t nw = 0;
        for (size_t w = 0; w < (size_t)grid * 4; ++w) {
            if (!h[w * 8 + 5]) continue;
            ++nw;
            for (int i = 0; i < 7; ++i) sm[i] += (double)h[w * 8 + i];
        }
        fprintf(stderr, "[w6 stamps] per-wave cycles: prologue+stem %.0f | barrier waits %.0f | epilogues %.0f | convolutions %.0f | heads %.0f | total %.0f | clock %.3f GHz\n",
                sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw, sm[5] / nw, sm[5] / sm[6] * 0.1);
        (void)hipFree(a.dbg);
    }
    return OTH_OK;
#endif
    hipLaunchKernelGGL(k_trunk_w6, dim3(grid), dim3(256), k6Lds, stream, a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

}  // namespace oth
