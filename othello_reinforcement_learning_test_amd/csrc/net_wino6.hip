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
// all nine N-tiles are full (48 = 3 x 16).  The transformed operand V lives in LDS in FRAGMENT ORDER (round 4):
// [lane group 3][k-group g4 4][j 3][xi 4][k-step 2][hi | lo][column c 16][16 B] = 144 KB for eight positions, so that a
// ds_read_b128 of (lane group, j, xi, k-step, half) reads four 256-byte runs, one per k-group, 12 KB apart: what the LDS
// serves at full rate (tools/probes/probe_w6_step.hip: the round-3 image -- tile x 3 KB with an XOR-swizzled slot, conflict-
// free by the lane-group model of the guide -- cost 4.9 cycles per MFMA in the convolution, 28 % of it in the kernel).  A row
// tap is a shift of the column by one (its 16-byte slot; column 0 / 15 take theirs from the neighbouring lane group's block);
// lanes whose source row is off the board read a 12 KB block of zeros.  Every weight is loaded once per EIGHT positions
// (k_trunk_h3: once per two).
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
using u32x4 = unsigned __attribute__((ext_vector_type(4)));   // a weight fragment as a register operand (HIP's uint4 is a struct)

constexpr int k6F = 64, k6BS = 6, k6Cells = 36, k6NP = 37, k6TP = 8;
constexpr int k6NT = 9;                         // N-tiles of a wave: 3 lane groups x 3 tile columns
constexpr int k6RunBytes = 256;                 // 16 columns x 16 B: what one k-group of a ds_read_b128 reads
constexpr int k6PlaneBytes = 3 * 4 * 2 * 2 * k6RunBytes;   // (j, xi, k-step, half) runs of one (lane group, k-group): 12288
constexpr int k6LgBytes = 4 * k6PlaneBytes;     // 49152
constexpr int k6VBytes = 3 * k6LgBytes;         // 147456
constexpr int k6ZeroOff = k6VBytes;             // a plane of zeros: the source of out-of-board rows (any j, xi, k-step, half)
constexpr int k6Lds = k6VBytes + k6PlaneBytes;  // 159744 of the CU's 163840
// byte offset of the run (j, xi, k-step, half) inside a plane
__host__ __device__ constexpr int k6_run(int j, int xi, int kk, int half) { return (((j * 4 + xi) * 2 + kk) * 2 + half) * k6RunBytes; }
constexpr float k6Clamp = 30000.0f;             // |V| <= 2 x (activation x act_scale) must stay in the f16 range: activations
                                                // <= 30000 / act_scale (1875 at the default 16; oth_net::act_scale)
constexpr int k6Groups = 6;                     // (row tap d, k-step kk of 32 input channels): g = 2*d + kk
constexpr int k6GroupU4 = 4 * 8 * 64;           // uint4 per group: 4 waves x 8 fragments x 64 lanes

struct Wino6Weights {
    int blocks = 0;
    uint4* d_w = nullptr;     // [layer][group 6][wave 4][xi 4][hi, lo][64 lanes] x 16 B   (A fragments of U)
    uint4* d_stem = nullptr;  // [wave 4][hi, lo][64 lanes] x 16 B: direct 3x3 stem as one k-step of 32 (27 used)
    float* d_bias = nullptr;  // [1 + 2*blocks][64], x act_scale (register_scaled_bias)
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
    float act_scale;           // oth_net::act_scale: the stem's input value and the heads' un-scaling
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

// The MFMA builtin: 36 accumulators + 18 residual registers + the weight ring need more than 256 architectural VGPRs; with
// the builtin the allocator keeps the accumulators in AGPRs (MFMA reads and writes them there) and knows the hazards.  (An
// in-place inline-asm form with VGPR accumulators and AGPR weight operands was measured in round 4 -- no faster, and one build of
// it miscompared with 102 hazards by tools/check_mfma_hazards.py: DESIGN_HISTORY.md K3d; removed from the sources in round 5.)
__device__ __forceinline__ f32x4 w6mfma(u32x4 a4, half8 b, f32x4 c) {
    const half8 a = __builtin_bit_cast(half8, a4);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 w6mfma0(u32x4 a4, half8 b) {   // first product of a chain: C = 0
    const half8 a = __builtin_bit_cast(half8, a4);
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
// at 440 / 400 registers (k6Regs = 220 / 200) the tree kernel is back beside it, but the spills cost more: 17.6 k / 17.0 k
// games/s on configs[4] against 23.1 k uncapped (k_trunk_h3: 21.9 k on the same box).
constexpr int k6Regs = 256;   // x 2 = total registers per lane (VGPR + AGPR)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(k6Regs))) void k_trunk_w6(Wino6Args a, const uint64_t* __restrict__ sb,
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
        const _Float16 one = (x & 1) ? (_Float16)(-a.act_scale) : (_Float16)a.act_scale;
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
    for (int i = tid; i < k6PlaneBytes / 16; i += 256) ((uint4*)(lds + k6ZeroOff))[i] = make_uint4(0, 0, 0, 0);   // the zero plane
    __syncthreads();

    f32x4 acc[4][k6NT];   // [xi][N-tile]
    f32x4 res[k6NT][2];   // [N-tile][x parity]: the residual in the spatial domain, fp32, x act_scale
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
                    acc[3 * e][lg * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh, acc[3 * e][lg * 3 + j], 0, 0, 0);
                    acc[3 * e][lg * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[3 * e][lg * 3 + j], 0, 0, 0);
                }
    }

    // V addressing (see the header): run (j, xi, k-step, half) of plane (lane group, k-group) at lg * 48 KB + g4 * 12 KB +
    // k6_run(...), column c at + 16 c.  A reader lane (c, g4) of lane group lg and row tap d takes column lg * 16 + c + d - 1
    // -- the same position's neighbouring row -- or the zero plane; j, xi, k-step and half are immediate offsets.  A writer
    // lane holds four consecutive channels (8 bytes of a 16-byte chunk): channels 16 wave + 4 g4 + r = chunk 2 wave + (g4 >> 1)
    // of the 64, i.e. k-step wave >> 1, k-group 2 (wave & 1) + (g4 >> 1), bytes 8 (g4 & 1) of its column's slot.
    const int ch0 = wave * 16 + 4 * g4;                       // + r: this lane's four output channels
    uint32_t wr_off[3];      // store address of (lane group; j = 0, xi = 0, hi half), this wave's k-step
    uint32_t rd_base[3][3];  // read address of (lane group, row tap): run 0 -- or the zero plane
#pragma unroll
    for (int lg = 0; lg < 3; ++lg) {
        wr_off[lg] = (uint32_t)(lg * k6LgBytes + (2 * (wave & 1) + (g4 >> 1)) * k6PlaneBytes + k6_run(0, 0, wave >> 1, 0) + c * 16 +
                                8 * (g4 & 1));
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int rs = row_l[lg] + d - 1;
            const int ps = pr_l[lg] + d - 1;   // (position, source row): same position when rs is on the board
            rd_base[lg][d] = (rs >= 0 && rs < k6BS) ? (uint32_t)((ps >> 4) * k6LgBytes + g4 * k6PlaneBytes + (ps & 15) * 16)
                                                    : (uint32_t)(k6ZeroOff + (ps & 15) * 16);   // its own slot: no bank shared
                                                                                                  // with a live lane of the k-group
        }
    }

    // [layer loop: begin]  (tools/probes/apply_ablation.py w6_exp_lgpipe swaps everything up to "[layer loop: end]" for the
    //                       lane-group pipeline of tools/probes/w6_lgpipe_loop.inc; comments only, the ISA is unchanged)
    const int n_layers = 1 + a.n_res_layers;
    uint32_t sat_bits = 0;
    OTH_W6STAMP(0)
    u32x4 wq[2][8];   // weight ring: [group parity][xi hi, xi lo]; a group = (row tap, k-step): 8 fragments
    float4 b4 = *(const float4*)(a.bias + ch0), b4n = b4;   // bias (x act_scale) and 1 / weight scale of the layer in the epilogue
    float inv = a.inv[0], invn = inv;
    for (int layer = 0; layer < n_layers; ++layer) {
        const bool last = layer == n_layers - 1;
        if (layer > 0) {   // loaded during the previous convolution
            b4 = b4n;
            inv = invn;
        }
        const f32x2 inv2 = {inv, inv};
        // A fragments of conv `layer+1`: group g at wl + g * k6GroupU4
        const u32x4* wl = (const u32x4*)a.w + (size_t)layer * (k6Groups * k6GroupU4) + (size_t)wave * (8 * 64) + lane;
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
                        uint2 hi[4], lo[4];
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi) {
                            hi[xi].x = wpack(V[xi][0].x, V[xi][0].y);
                            hi[xi].y = wpack(V[xi][1].x, V[xi][1].y);
                            lo[xi].x = wresid(hi[xi].x, V[xi][0].x, V[xi][0].y);
                            lo[xi].y = wresid(hi[xi].y, V[xi][1].x, V[xi][1].y);
                        }
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi) {
                            *(uint2*)(lds + wr_off[lg] + k6_run(j, xi, 0, 0)) = hi[xi];
                            *(uint2*)(lds + wr_off[lg] + k6_run(j, xi, 0, 1)) = lo[xi];
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
        auto src_of = [&](int q, int half) -> uint32_t {
            const int xi = q & 3, nt = (q >> 2) % k6NT, grp = q / GS, kk = grp & 1, d = grp >> 1;
            return rd_base[nt / 3][d] + (uint32_t)k6_run(nt % 3, xi, kk, half);
        };
        constexpr int PD = 2;   // LDS operand pairs in flight ahead of the MFMAs (steps)
        half8 xh[PD + 1], xl[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            xh[q] = *(const half8*)(lds + src_of(q, 0));
            xl[q] = *(const half8*)(lds + src_of(q, 1));
        }
        auto conv_d = [&](auto DC) {
            constexpr int D = decltype(DC)::value;
#pragma unroll
            for (int ql = 0; ql < 2 * GS; ++ql) {
                const int q = D * 2 * GS + ql;
                const int xi = q & 3, nt = (q >> 2) % k6NT, grp = q / GS, sl = q % (PD + 1), psl = (q + PD) % (PD + 1);
                const int step = q % GS;
                const u32x4 wh = wq[grp & 1][2 * xi], wlo = wq[grp & 1][2 * xi + 1];
                OTH_W6SB;
                if (q < GS) acc[xi][nt] = w6mfma0(wh, xl[sl]);     // the layer's first group starts every accumulator
                else acc[xi][nt] = w6mfma(wh, xl[sl], acc[xi][nt]);
                OTH_W6SB;
                if (q + PD < QT) xh[psl] = *(const half8*)(lds + src_of(q + PD, 0));
                OTH_W6SB;
                acc[xi][nt] = w6mfma(wh, xh[sl], acc[xi][nt]);
                OTH_W6SB;
                if (q + PD < QT) xl[psl] = *(const half8*)(lds + src_of(q + PD, 1));
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
    // [layer loop: end]

    // ---------------- heads (fp32 VALU): final activations (in `res`, x act_scale) -> LDS planes [channel][8 x 36 cells] f32
    //                  (aliasing V: every read of it is done), then each wave runs the shared one-wave head code on two
    //                  of the eight positions
    if (sat_bits >= __float_as_uint(k6Clamp)) atomicOr(a.sat, 1);
    __syncthreads();
    constexpr int NCO = k6TP * k6Cells;   // 288
    float* planes = (float*)lds;
    const float us = 1.0f / a.act_scale;
#pragma unroll
    for (int lg = 0; lg < 3; ++lg)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ci = p_l[lg] * k6Cells + row_l[lg] * k6BS + 2 * j + e;
                const f32x4 v = res[lg * 3 + j][e];
#pragma unroll
                for (int r = 0; r < 4; ++r) planes[(ch0 + r) * NCO + ci] = v[r] * us;
            }
    __syncthreads();
    // [heads: begin]
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
        heads_wave_n<k6F, k6BS, 2, true>(a.heads, a.pfc_wt, a.vfc1_wt, srcs, NCO, scratch, lane, lps, vs, live);
    }
    // [heads: end]
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
        for (int i = 0; i < k6F; ++i) bias[i] = cv.bias[i];
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
        for (int i = 0; i < k6F; ++i) bias[(size_t)(li + 1) * k6F + i] = cv.bias[i];
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
    if (int rc = register_scaled_bias(net, ww->d_bias, std::move(bias))) return rc;
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
    a.act_scale = net->act_scale;
    static bool attr_set_dev[64] = {};
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk_w6, hipFuncAttributeMaxDynamicSharedMemorySize, k6Lds));
        attr_set = true;
    }
    const unsigned grid = (unsigned)((n + k6TP - 1) / k6TP);
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
