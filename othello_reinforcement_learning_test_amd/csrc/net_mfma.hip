// net_mfma.hip -- OthelloResNet forward for 128 filters as ONE fused gfx950 kernel.
//
// Reference: /root/reference/src/model/net.py:182-205 (eval mode; BatchNorm folded at load time).
//
//   k_trunk16<X3, TP>  v_mfma_f32_16x16x32_f16, TP = 2 positions per workgroup, two workgroups per CU, tiles = board row
//                      of a position pair (padding rows skipped).  TP = 1 (tile = two rows of one position) serves
//                      launches of <= 256 positions: half the latency when every workgroup has a CU to itself anyway;
//                      TP = 4 is the single-pass f16 precision's build (its TP = 2 instantiation spills registers).
//   (The first version of this kernel -- v_mfma_f32_32x32x16_f16, four positions per workgroup, OTH_MFMA_SHAPE=32 -- was
//   removed in round 5: slower since round 1, DESIGN_HISTORY.md section 7.1; it is in the history at 274f240.)
//
// Design (MI355X-first):
//   * a workgroup of 4 waves carries its positions through the stem, all residual blocks and both heads in one
//     launch; the activations NEVER leave the CU: LDS [cell][hi 128 x f16 | lo 128 x f16] (512 B per cell), every
//     3x3 tap reads them in place (implicit GEMM, no im2col copy); out-of-board taps read a zeroed cell.
//   * wave w owns output channels [32w, 32w+32): its weight operand streams from L2 straight into registers in
//     MFMA fragment order (host-packed, 1 KiB per wave-load, every byte loaded exactly once per workgroup and
//     layer); accumulators and the fp32 residual stay in registers, so there is no barrier inside a layer -- two
//     LDS-only barriers per layer around the in-place activation rewrite.
//   * arithmetic: both operands split a = a_hi + a_lo (f16 pairs, ~22 significant bits), three products
//     a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in fp32: fp32-equivalent results (a single f16 pass is 1e-2
//     off on trained-like weights).  OTH_PREC_F16 runs the same kernels with the hi parts only.
//   * power-of-two scaling of activations (2^4, carried through the residual) and of each layer's weights keeps the
//     lo parts in the normal f16 range; it is undone exactly on the fp32 accumulator.
//   * LDS bank conflicts: the 16-byte chunk index of a cell is XORed with a key derived from the source cell so that
//     the 16 lanes of every ds_read_b128 group hit 16 different slots for all nine taps (tools/lds_bank_model.py).
//   * the chip is power/clock-limited on this code (76 % matrix-pipe busy at ~1.76 of 2.4 GHz), so instruction ORDER
//     matters: the three split products of an accumulator issue back to back with one operand changing at a time
//     (k_trunk16's tile body): same instructions, -4 % time.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "net.h"
#include "net_heads.h"

namespace oth {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using half4 = _Float16 __attribute__((ext_vector_type(4)));

constexpr int kCellBytes = 512;                // 256 B hi + 256 B lo
constexpr int kActBytes = 256 * kCellBytes;    // 131072

constexpr int kScratchOff = kActBytes + 512;   // stem im2col (16 KiB) / head scratch
constexpr int kLdsBytes = kScratchOff + 16384;
constexpr float kActClamp = 60000.0f;          // activations x act_scale are clamped to the f16 range of the hi parts: activations
                                               // <= 60000 / act_scale (3750 at the default scale 16; oth_net::act_scale)

struct MfmaWeights {
    int blocks = 0;
    uint4* d_w = nullptr;      // fragments: [layer][step 0..71][wave 0..3][plane hi,lo][64 lanes] x 16 B
    uint4* d_stem = nullptr;   // [step 0..1][wave][plane][64 lanes]
    float* d_bias = nullptr;   // [1 + 2*blocks][128], x act_scale (register_scaled_bias)
    float* d_inv = nullptr;    // [1 + 2*blocks] 1 / weight_scale
};

struct MfmaArgs {
    const uint4* w;
    const uint4* stem;
    const float* bias;
    const float* inv;
    int n_res_layers;  // 2 * blocks
    HeadParams heads;
    unsigned long long* dbg;  // diagnostic build (-DOTH_STAMPS) only: per-wave phase cycle sums
    unsigned long long* tl;   // diagnostic build only: per-layer conv start / end of workgroups 0 and 256
    int* sat;                 // set to 1 when an activation reaches the clamp (oth_net_saturated)
    float act_scale;          // oth_net::act_scale: the stem's input value and the heads' un-scaling
};

#ifdef OTH_STAMPS
// Diagnostic build: s_memtime phase stamps (never compiled into the shipped library).
__device__ __forceinline__ unsigned long long oth_clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long oth_realclk() {   // 100 MHz constant clock
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define OTH_STAMP(i) { const unsigned long long t1_ = oth_clk(); ph_[i] += t1_ - t0_; t0_ = t1_; }
#else
#define OTH_STAMP(i)
#endif

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt(0), which would
// expose the L2 latency of the weight fragments prefetched across the layer boundary.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// =================================================================================================
// k_trunk16: v_mfma_f32_16x16x32_f16, TP positions per workgroup.
// Per wave: 2 row blocks of 16 output channels x NT tiles of 16 cells; one k-step = 32 input channels.
//   A operand (weights):     lane l holds W[row = l&15][k = 8*(l>>4) + j]
//   B operand (activations): lane l holds X[k = 8*(l>>4) + j][col = l&15 = cell of the tile]
//   D: lane l holds column l&15 (cell), rows 4*(l>>4) + reg: 4 consecutive channels -> 8-byte stores
// A tile is BOARD ROW y OF A PAIR OF POSITIONS (lane c16: position c16>>3 of the pair, column c16&7), not two
// rows of one position: for a tap with dy = -1 (+1) the tile of row 0 (7) reads only padding, so its LDS
// reads and MFMAs are skipped altogether -- 1/12 of the conv work (zero padding is 16 % of a 3x3 conv on 8x8).
// =================================================================================================
using f32x4 = float __attribute__((ext_vector_type(4)));
template <bool INPLACE = true>
__device__ __forceinline__ f32x4 mfma32(half8 a, half8 b, f32x4 c) {
    if constexpr (!INPLACE) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    // accumulate IN PLACE (vDst = SrcC): the register allocator otherwise moves an accumulator to fresh registers at
    // the head of a chain and pays WAR wait states (s_nop) when the freed registers are reused at once
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
// Chunk swizzle: within a ds_read_b128 lane group the two chunk indices differ in bit 0 and the cells split as
// {(pos 0, x lo), (pos 1, x hi)} vs {(pos 0, x hi), (pos 1, x lo)} (x lo/hi = the two halves of the 8 shifted
// columns), so key bit 0 must flip exactly when BOTH the position bit and the x half flip:
// key = p | x[1:0] << 1 | (x[2] ^ p) << 3.  tools/lds_bank_model.py: 4.0 cycles per read for all taps.
__device__ __forceinline__ int swz16(int xs, int p) {
    const int x = xs & 7;
    return (p & 1) | ((x & 3) << 1) | ((((x >> 2) & 1) ^ (p & 1)) << 3);
}
// i-th tile that has in-board source rows for a tap with row offset DY (tile t <-> board row t & 7)
template <int DY>
__device__ __forceinline__ constexpr int valid_tile(int i) {
    return DY == 0 ? i : (DY < 0 ? i + 1 + (i >= 7 ? 1 : 0) : i + (i >= 7 ? 1 : 0));
}

template <bool X3, int TP>
__global__ __launch_bounds__(256, (TP == 2 ? 2 : 1)) void k_trunk16(MfmaArgs a, const uint64_t* __restrict__ sb,
                                                                   const uint64_t* __restrict__ ob,
                                                                   const uint64_t* __restrict__ lgl, int64_t n,
                                                                   const int32_t* __restrict__ n_valid,
                                                                   float* __restrict__ logp, float* __restrict__ vout) {
    // in-place asm MFMAs only in the build that gains from them and that tools/check_mfma_hazards.py finds clean (TP = 2):
    // the 4-position build keeps some accumulators in AGPRs and copies MFMA results there at once, which needs the
    // compiler's own wait states
    constexpr bool IP = TP == 2;
    // the interleaved issue order pays where two workgroups share a CU and saturate the matrix pipe (TP = 2: -3 %); the
    // low-latency one-position build (launches <= 256 positions, a CU per workgroup) is 2-4 % faster with round 2's
    // clumped order and builtin MFMAs (same-session A/B, profiles/r03_trunk_experiments.log), so it keeps them
    constexpr bool ILV = TP == 2;
    constexpr int PD = (X3 || TP <= 2) ? 2 : 4;  // activation fragments in flight (tiles of 16 cells)
    // weight k-steps (32 channels) in flight: a k-step is NT tiles x 6 MFMAs, i.e. 768 cycles at TP = 2 (one ahead covers
    // the L2 latency; two spill registers) but only 384 at TP = 1, where one ahead stalls every k-step (small engines
    // lost 25-30 % with it) and registers are plentiful
    constexpr int PB = TP == 1 ? 2 : 1;   // (TP = 2: one k-step -- 8 tiles x 6 MFMAs = 768+ cycles -- ahead covers the L2 latency; 2 spills registers)
    constexpr int NT = 4 * TP;      // 16-cell tiles per workgroup: (pair of positions) x (board row); TP = 1: two rows
    constexpr int ZERO_OFF = TP * 64 * kCellBytes;   // zero cell (512 B) after the activations
    constexpr int SCR_OFF = ZERO_OFF + 512;          // stem im2col (TP*4 KiB) / head scratch
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int64_t pos0 = (int64_t)blockIdx.x * TP;
    if (pos0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, c16 = lane & 15;
    const int pz = c16 >> 3, cx = c16 & 7;   // position within the pair (TP = 1: row parity), board column
    // LDS cell index of this lane in tile t: ((2*(t>>3) + pz)*64 + (t&7)*8 + cx); lane part and tile part.
    // TP = 1 (low-latency build for launches that do not fill the chip): a tile is rows 2t, 2t+1 of the one
    // position, cell = 16 t + 8 pz + cx; the swizzle key then uses the row parity where the pair build uses the
    // position bit, which keeps the lane groups of every ds_read_b128 on 16 different chunks in the same way.
    const uint32_t lane_cell = TP == 1 ? (uint32_t)c16 : (uint32_t)(pz * 64 + cx);
#define OTH_TILE_CELL(t) (TP == 1 ? (uint32_t)((t) * 16) : (uint32_t)(((t) >> 3) * 128 + ((t) & 7) * 8))

    if (tid < TP * 64) {  // stem input: im2col of the three bit planes, [TP*64 cells][32 k] f16, k = tap*3 + plane
        const int p = tid >> 6, c = tid & 63, y = c >> 3, x = c & 7;
        const bool live = pos0 + p < nv;
        const uint64_t b0 = live ? sb[pos0 + p] : 0, b1 = live ? ob[pos0 + p] : 0, b2 = live ? lgl[pos0 + p] : 0;
        _Float16 vals[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) vals[i] = (_Float16)0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
            const int s = ok ? yy * 8 + xx : 0;
            vals[tap * 3 + 0] = (ok && ((b0 >> s) & 1ULL)) ? (_Float16)a.act_scale : (_Float16)0.0f;
            vals[tap * 3 + 1] = (ok && ((b1 >> s) & 1ULL)) ? (_Float16)a.act_scale : (_Float16)0.0f;
            vals[tap * 3 + 2] = (ok && ((b2 >> s) & 1ULL)) ? (_Float16)a.act_scale : (_Float16)0.0f;
        }
        half8* dst = (half8*)(lds + SCR_OFF + tid * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = vals[q * 8 + j];
            dst[q] = t;
        }
    }
    if (tid < 32) ((uint4*)(lds + ZERO_OFF))[tid] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    f32x4 acc[NT][2], res[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][rb][i] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) OTH_PIN_ACC(acc[t][rb]);
    OTH_PIN_ACC_END();

    {   // stem conv: one k-step of 32
        const uint4* wp = a.stem + ((size_t)wave * 4) * 64 + lane;  // [rb0 hi][rb0 lo][rb1 hi][rb1 lo]
        half8 wh[2], wlo[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            wh[rb] = __builtin_bit_cast(half8, wp[(rb * 2) * 64]);
            wlo[rb] = __builtin_bit_cast(half8, wp[(rb * 2 + 1) * 64]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const half8 xh = *(const half8*)(lds + SCR_OFF + (OTH_TILE_CELL(t) + lane_cell) * 64 + g4 * 16);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if (X3) acc[t][rb] = mfma32<IP>(wlo[rb], xh, acc[t][rb]);
                acc[t][rb] = mfma32<IP>(wh[rb], xh, acc[t][rb]);
            }
        }
    }

    const uint32_t keyw = (uint32_t)swz16(cx, pz);
    uint32_t wr_off[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
        wr_off[rb] = lane_cell * kCellBytes + ((((uint32_t)(wave * 4 + rb * 2 + (g4 >> 1))) ^ keyw) << 4) +
                     8u * (uint32_t)(g4 & 1);
    const int ch0 = wave * 32 + 4 * g4;  // + 16*rb + e

    const int n_layers = 1 + a.n_res_layers;
    uint4 wq[PB][4];  // weight ring [slot][rb0 hi, rb0 lo, rb1 hi, rb1 lo] of the conv that follows
    uint32_t sat_bits = 0;   // largest activation seen (bit pattern): reaching the clamp sets the network's saturation flag
#ifdef OTH_STAMPS
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, t0_ = oth_clk(), tstart_ = t0_;
    const unsigned long long rstart_ = oth_realclk();
#endif
    for (int layer = 0; layer < n_layers; ++layer) {
        const bool last = layer == n_layers - 1;
        // bias first, weights after: the first use of b4 then waits with vmcnt(#weight loads), not vmcnt(0)
        float4 b4[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) b4[rb] = *(const float4*)(a.bias + layer * 128 + ch0 + 16 * rb);
        const float inv = a.inv[layer];
        const uint4* wl = a.w + ((size_t)layer * 36 * 4 + wave) * 256 + lane;  // + step*1024 + frag*64
        if (!last) {  // first weight fragments of conv `layer+1`: their L2 latency hides under the epilogue
#pragma unroll
            for (int i = 0; i < PB; ++i)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    if (X3 || !(f & 1)) wq[i][f] = wl[(size_t)i * 1024 + f * 64];
        }
        // ---------------- epilogue of conv `layer` (0 = stem): scale back, bias, skip, ReLU, re-split.  The three
        // flags are compile-time (four instantiations of the same body): a run-time add_res / set_res costs two
        // v_cndmask per value and a branch per store on `last` (a fifth of the epilogue's VALU work).
        auto epilogue = [&](auto ADD, auto SET, auto LAST) {
            constexpr bool add_res = decltype(ADD)::value;    // second conv of a block (net.py:58)
            constexpr bool set_res = decltype(SET)::value;
            constexpr bool is_last = decltype(LAST)::value;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    float vs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bb = e == 0 ? b4[rb].x : (e == 1 ? b4[rb].y : (e == 2 ? b4[rb].z : b4[rb].w));
                        float v = fmaf(acc[t][rb][e], inv, bb);
                        if (add_res) v += res[t][rb][e];
                        v = __builtin_amdgcn_fmed3f(v, 0.f, kActClamp);  // ReLU + f16 range clamp (x <= 3750)
                        if (set_res) res[t][rb][e] = v;
                        vs[e] = v;
                    }
                    // saturation watch: non-negative floats order like their bit patterns (one v_max3_u32 per pair)
                    sat_bits = max(sat_bits, max(__float_as_uint(vs[0]), __float_as_uint(vs[1])));
                    sat_bits = max(sat_bits, max(__float_as_uint(vs[2]), __float_as_uint(vs[3])));
                    if (!is_last) {
                        half4 hi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hi[e] = (_Float16)vs[e];
                        char* dst = lds + OTH_TILE_CELL(t) * kCellBytes + wr_off[rb];
                        *(half4*)dst = hi;
                        if (X3) {
                            half4 lo;
#pragma unroll
                            for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(vs[e] - (float)hi[e]);
                            *(half4*)(dst + 256) = lo;
                        }
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        OTH_STAMP(0)
        lds_barrier();  // every wave has finished reading the previous activations
        OTH_STAMP(1)
        if (layer == 0) {
            if (last) epilogue(F_{}, T_{}, T_{});
            else epilogue(F_{}, T_{}, F_{});
        } else if (last) {
            epilogue(T_{}, T_{}, T_{});      // the last layer is the second conv of the last block
        } else if (layer & 1) {
            epilogue(F_{}, F_{}, F_{});
        } else {
            epilogue(T_{}, T_{}, F_{});
        }
        if (last) break;
        OTH_STAMP(2)
        lds_barrier();
        OTH_STAMP(3)
        // zero the accumulators HERE (they need no registers during the epilogue) and keep the moves in front of the
        // wait states the in-place asm MFMAs need after a VALU write (net.h)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                OTH_PIN_ACC(acc[t][rb]);
            }
        OTH_PIN_ACC_END();
#ifdef OTH_STAMPS
        if (a.tl && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256) && layer < 21)
            a.tl[(blockIdx.x ? 64 : 0) + layer * 2] = t0_;
#endif

        // ---------------- conv `layer+1`: 3 row offsets (compile-time) x 3 column offsets x 4 k-steps x tiles
        auto tap_row = [&](auto DYC) {
            constexpr int DY = decltype(DYC)::value;
            constexpr int NTV = (DY == 0 || TP == 1) ? NT : NT - NT / 8;  // tiles with in-board source rows
            constexpr int NQ = 4 * NTV;                      // (k-step, tile) fragments of one tap
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int tap = (DY + 1) * 3 + dxi;
                const int xs = cx + dxi - 1;
                const bool xok = xs >= 0 && xs < 8;
                const uint32_t hk = (uint32_t)(g4 ^ swz16(xs, TP == 1 ? (pz ^ (DY & 1)) : pz));
                // this lane's source cell of tile t.  Pair build: row (t&7)+DY is in-board for every tile used here.
                // TP = 1: row 2t+pz+DY leaves the board only for (DY = -1, t = 0, pz = 0) and (DY = +1, t = 3, pz = 1).
                const int src_cell = (TP == 1 ? pz * 8 : pz * 64) + DY * 8 + xs;   // + OTH_TILE_CELL(t)
#define OTH_TILE_OF(i) (TP == 1 ? (i) % NTV : valid_tile<DY>((i) % NTV))
#define OTH_ROW_OK(t) (TP != 1 || DY == 0 || (DY < 0 ? ((t) > 0 || pz != 0) : ((t) < 3 || pz == 0)))
#define OTH_SRC(i)                                                                                                   \
    (lds + (((xok && OTH_ROW_OK(OTH_TILE_OF(i))) ? (uint32_t)((int)OTH_TILE_CELL(OTH_TILE_OF(i)) + src_cell) * kCellBytes : (uint32_t)ZERO_OFF) | \
            ((((uint32_t)(((i) / NTV) << 2)) ^ hk) << 4)))
                half8 xh[PD + 1], xl[PD + 1];
#pragma unroll
                for (int q = 0; q < PD; ++q) {
                    xh[q] = *(const half8*)OTH_SRC(q);
                    if (X3) xl[q] = *(const half8*)(OTH_SRC(q) + 256);
                }
                half8 wh[2], wlo[2];
                if constexpr (ILV) {
                // Interleaved issue order (every statement pinned by sched_barrier): the address arithmetic and the two
                // LDS reads of tile q+PD and, at a k-step boundary, the four weight loads of k-step +PB are placed BETWEEN
                // the six MFMAs of tile q -- one per MFMA slot (an MFMA holds the issue port for 8 of its 16 cycles) --
                // instead of in a clump between two tiles, where they leave the matrix pipe idle for a wave that has
                // the SIMD to itself (a lone wave ran 22.6 cycles per MFMA with the clumped order).
#define OTH_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const bool pf = q + PD < NQ;
                    const int psl = (q + PD) % (PD + 1);
                    const bool neww = (q % NTV) == 0;
                    const int wslot = (q / NTV) % PB;
                    int nstep = tap * 4 + q / NTV + PB;
                    nstep = nstep < 36 ? nstep : 35;
                    if (neww) {
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            wh[rb] = __builtin_bit_cast(half8, wq[wslot][rb * 2]);
                            if (X3) wlo[rb] = __builtin_bit_cast(half8, wq[wslot][rb * 2 + 1]);
                        }
                    }
                    const int t = OTH_TILE_OF(q);
                    const int sl = q % (PD + 1);
                    OTH_SB;
                    if constexpr (X3) {
                        acc[t][0] = mfma32<IP>(wh[0], xl[sl], acc[t][0]);
                        OTH_SB;
                        const char* src = pf ? OTH_SRC(q + PD) : lds;
                        OTH_SB;
                        acc[t][0] = mfma32<IP>(wh[0], xh[sl], acc[t][0]);
                        OTH_SB;
                        if (neww) wq[wslot][0] = wl[(size_t)nstep * 1024];
                        OTH_SB;
                        acc[t][0] = mfma32<IP>(wlo[0], xh[sl], acc[t][0]);
                        OTH_SB;
                        if (pf) xh[psl] = *(const half8*)src;
                        if (neww) wq[wslot][1] = wl[(size_t)nstep * 1024 + 64];
                        OTH_SB;
                        acc[t][1] = mfma32<IP>(wlo[1], xh[sl], acc[t][1]);
                        OTH_SB;
                        if (pf) xl[psl] = *(const half8*)(src + 256);
                        if (neww) wq[wslot][2] = wl[(size_t)nstep * 1024 + 128];
                        OTH_SB;
                        acc[t][1] = mfma32<IP>(wh[1], xh[sl], acc[t][1]);
                        OTH_SB;
                        if (neww) wq[wslot][3] = wl[(size_t)nstep * 1024 + 192];
                        OTH_SB;
                        acc[t][1] = mfma32<IP>(wh[1], xl[sl], acc[t][1]);
                    } else {
                        if (pf) xh[psl] = *(const half8*)OTH_SRC(q + PD);
                        if (neww) {
                            wq[wslot][0] = wl[(size_t)nstep * 1024];
                            wq[wslot][2] = wl[(size_t)nstep * 1024 + 128];
                        }
                        OTH_SB;
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) acc[t][rb] = mfma32<IP>(wh[rb], xh[sl], acc[t][rb]);
                    }
                    OTH_SB;
                }
                } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + PD < NQ) {
                        const int slot = (q + PD) % (PD + 1);
                        xh[slot] = *(const half8*)OTH_SRC(q + PD);
                        if (X3) xl[slot] = *(const half8*)(OTH_SRC(q + PD) + 256);
                    }
                    if ((q % NTV) == 0) {  // new k-step of 32 channels
                        const int kk = q / NTV, slot = kk % PB;
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            wh[rb] = __builtin_bit_cast(half8, wq[slot][rb * 2]);
                            if (X3) wlo[rb] = __builtin_bit_cast(half8, wq[slot][rb * 2 + 1]);
                        }
                        int nstep = tap * 4 + kk + PB;
                        nstep = nstep < 36 ? nstep : 35;
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            if (X3 || !(f & 1)) wq[slot][f] = wl[(size_t)nstep * 1024 + f * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int t = OTH_TILE_OF(q);
                    if constexpr (X3) {
                        // MFMA order inside a tile (measured: -4 % kernel time, +6 % games/s on the whole bench -- the chip
                        // is power-limited and this order switches less per instruction): the three products of one
                        // accumulator run back to back (result forwarded instead of re-read) and exactly one of A / B
                        // changes from one MFMA to the next: A = wh0 wh0 wlo0 | wlo1 wh1 wh1, B = xl xh xh | xh xh xl.
                        // (Alternating the two row blocks from tile to tile would keep the weight operand across the tile
                        // boundary too, +0.4 %, but makes the summation order of a cell depend on its tile index and so
                        // on the positions-per-workgroup build: results would no longer be bit-identical across batch
                        // sizes.  Each row block therefore always sums in the same order.)
                        const int sl = q % (PD + 1);
                        const int r0 = 0, r1 = 1;
                        acc[t][r0] = mfma32<IP>(wh[r0], xl[sl], acc[t][r0]);
                        acc[t][r0] = mfma32<IP>(wh[r0], xh[sl], acc[t][r0]);
                        acc[t][r0] = mfma32<IP>(wlo[r0], xh[sl], acc[t][r0]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[t][r1] = mfma32<IP>(wlo[r1], xh[sl], acc[t][r1]);
                        acc[t][r1] = mfma32<IP>(wh[r1], xh[sl], acc[t][r1]);
                        acc[t][r1] = mfma32<IP>(wh[r1], xl[sl], acc[t][r1]);
                    } else {
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) acc[t][rb] = mfma32<IP>(wh[rb], xh[q % (PD + 1)], acc[t][rb]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
#undef OTH_SB
#undef OTH_SRC
#undef OTH_ROW_OK
#undef OTH_TILE_OF
            }
        };
        tap_row(std::integral_constant<int, -1>{});
        tap_row(std::integral_constant<int, 0>{});
        tap_row(std::integral_constant<int, 1>{});
        OTH_STAMP(4)
#ifdef OTH_STAMPS
        if (a.tl && lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 256) && layer < 21)
            a.tl[(blockIdx.x ? 64 : 0) + layer * 2 + 1] = t0_;
#endif
    }
#ifdef OTH_STAMPS
    if (a.dbg && lane == 0) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = ph_[i];
        o[5] = oth_clk() - tstart_;
        o[6] = oth_realclk() - rstart_;
        o[7] = rstart_;
    }
#endif

    if (sat_bits >= __float_as_uint(kActClamp)) atomicOr(a.sat, 1);   // rare: surfaced by oth_net_saturated
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float us = 1.0f / a.act_scale;
            const float4 o = make_float4(res[t][rb][0] * us, res[t][rb][1] * us, res[t][rb][2] * us, res[t][rb][3] * us);
            *(float4*)(lds + (size_t)(OTH_TILE_CELL(t) + lane_cell) * 512 + (size_t)(ch0 + 16 * rb) * 4) = o;
        }
    __syncthreads();
    for (int p = 0; p < TP; ++p) {
        if (pos0 + p >= nv) break;
        heads_forward(a.heads, 128, (const float*)(lds + (size_t)p * 64 * 512), 128, (float*)(lds + SCR_OFF),
                      logp + (pos0 + p) * 65, vout + pos0 + p);
    }
#undef OTH_TILE_CELL
}

// ------------------------------------------------------------------------------------------------
// host: pack weights into fragment order
// ------------------------------------------------------------------------------------------------
static inline void split_f16(float v, uint16_t& hi, uint16_t& lo) {
    const _Float16 hh = (_Float16)v;
    const _Float16 ll = (_Float16)(v - (float)hh);
    memcpy(&hi, &hh, 2);
    memcpy(&lo, &ll, 2);
}

// A fragment of v_mfma_f32_16x16x32_f16: lane l holds W[row = l&15][k = 8*(l>>4) + j].  Stream per layer:
// [step (32 channels)][wave][rb0 hi, rb0 lo, rb1 hi, rb1 lo][64 lanes] x 16 B.
static float pack_conv16(const FoldedConv& c, std::vector<uint16_t>& out, size_t base, int steps,
                         const std::vector<int>& k_map) {
    float mx = 0.f;
    for (float x : c.w) mx = fmaxf(mx, fabsf(x));
    int e = 0;
    if (mx > 0.f) e = (int)floorf(log2f(16384.0f / mx));
    if (e > 24) e = 24;
    if (e < -24) e = -24;
    const float scale = ldexpf(1.0f, e);
    for (int s = 0; s < steps; ++s)
        for (int w = 0; w < 4; ++w)
            for (int rb = 0; rb < 2; ++rb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int k = s * 32 + 8 * (l >> 4) + j, col = w * 32 + rb * 16 + (l & 15);
                        const int src = k_map[k];
                        const float v = src < 0 ? 0.f : c.w[(size_t)src * c.cout + col] * scale;
                        uint16_t hi, lo;
                        split_f16(v, hi, lo);
                        const size_t frag = base + ((size_t)(s * 4 + w) * 4 + rb * 2) * 64 * 8;  // in halfs
                        out[frag + (size_t)l * 8 + j] = hi;
                        out[frag + 64 * 8 + (size_t)l * 8 + j] = lo;
                    }
    return scale;
}

void mfma_free_weights(oth_net* net) {
    if (!net->mfma) return;
    if (net->mfma->d_w) (void)hipFree(net->mfma->d_w);
    if (net->mfma->d_stem) (void)hipFree(net->mfma->d_stem);
    if (net->mfma->d_bias) (void)hipFree(net->mfma->d_bias);
    if (net->mfma->d_inv) (void)hipFree(net->mfma->d_inv);
    delete net->mfma;
    net->mfma = nullptr;
}

int mfma_pack_weights(oth_net* net, int precision) {
    (void)precision;
    const HostNet& hn = net->host;
    OTH_CHECK(hn.filters == 128, "MFMA trunk needs 128 filters");
    const int L = 2 * hn.blocks;
    MfmaWeights* mw = new MfmaWeights();
    mw->blocks = hn.blocks;
    net->mfma = mw;
    const size_t frag_halfs = 2 * 64 * 8;                 // hi + lo, per (step, wave)
    const size_t layer_halfs = (size_t)72 * 4 * frag_halfs;
    std::vector<uint16_t> w((size_t)L * layer_halfs), stem((size_t)2 * 4 * frag_halfs);
    std::vector<float> bias((size_t)(L + 1) * 128), inv(L + 1);
    std::vector<int> km(1152);
    for (int k = 0; k < 1152; ++k) km[k] = k;  // k = tap*128 + ci, steps ordered (tap, kk)
    std::vector<int> ks(32, -1);               // stem: gemm k = tap*3 + plane for k < 27
    for (int k = 0; k < 27; ++k) ks[k] = k;
    {
        const float sc = pack_conv16(hn.stem, stem, 0, 1, ks);
        inv[0] = 1.0f / sc;  // accumulator holds (16 x) * (sc w): divide by sc to get 16 * y
        for (int i = 0; i < 128; ++i) bias[i] = hn.stem.bias[i];
    }
    for (int l = 0; l < L; ++l) {
        const float sc = pack_conv16(hn.res[l], w, (size_t)l * layer_halfs, 36, km);
        inv[l + 1] = 1.0f / sc;
        for (int i = 0; i < 128; ++i) bias[(size_t)(l + 1) * 128 + i] = hn.res[l].bias[i];
    }
    OTH_HIP(hipMalloc(&mw->d_w, w.size() * 2));
    OTH_HIP(hipMalloc(&mw->d_stem, stem.size() * 2));
    OTH_HIP(hipMalloc(&mw->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&mw->d_inv, inv.size() * 4));
    OTH_HIP(hipMemcpy(mw->d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(mw->d_stem, stem.data(), stem.size() * 2, hipMemcpyHostToDevice));
    if (int rc = register_scaled_bias(net, mw->d_bias, std::move(bias))) return rc;
    OTH_HIP(hipMemcpy(mw->d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    return OTH_OK;
}

int mfma_forward(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
                 const int32_t* n_valid, float* logp, float* v, hipStream_t stream) {
    OTH_CHECK(net->mfma, "MFMA weights not packed");
    MfmaArgs a;
    a.w = net->mfma->d_w;
    a.stem = net->mfma->d_stem;
    a.bias = net->mfma->d_bias;
    a.inv = net->mfma->d_inv;
    a.n_res_layers = 2 * net->mfma->blocks;
    a.heads = net->heads;
    a.dbg = nullptr;
    a.tl = nullptr;
    a.sat = net->d_sat;
    a.act_scale = net->act_scale;
#ifdef OTH_STAMPS
    OTH_HIP(hipMalloc(&a.tl, 128 * sizeof(unsigned long long)));
    OTH_HIP(hipMemset(a.tl, 0, 128 * sizeof(unsigned long long)));
    const unsigned dbg_grid = (unsigned)((n + 1) / 2);
    OTH_HIP(hipMalloc(&a.dbg, (size_t)dbg_grid * 4 * 8 * sizeof(unsigned long long)));
    OTH_HIP(hipMemset(a.dbg, 0, (size_t)dbg_grid * 4 * 8 * sizeof(unsigned long long)));
#endif
    static bool attr_set_dev[64] = {};  // per device: the attribute belongs to the (function, device) pair
    bool& attr_set = attr_set_dev[net->device & 63];
    constexpr int kLds2 = 2 * 64 * kCellBytes + 512 + 2 * 4096;  // TP = 2: 74 240 B, two workgroups per CU
    constexpr int kLds1 = 1 * 64 * kCellBytes + 512 + 2 * 4096;  // TP = 1 (heads scratch needs the 8 KiB)
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk16<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk16<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds1));
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk16<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2));
        attr_set = true;
    }
    const bool x3 = net->precision == OTH_PREC_F16X3 || net->precision == OTH_PREC_F16X3_DIRECT;
    // fp16x3: two 2-position workgroups per CU; launches that cannot fill the chip (n <= 256: every workgroup has a CU to
    // itself either way) run one position per workgroup -- twice the CUs, half the MFMAs per workgroup: 0.34 -> ~0.2 ms per
    // launch.  OTH_TRUNK_TP=1|2 (read per call) forces a build: the switch the variants test uses to compare the two on one
    // batch.  The single-pass f16 precision runs four positions per workgroup (its TP = 2 instantiation spills registers).
    const char* tpe = getenv("OTH_TRUNK_TP");
    const int tp_env = tpe ? atoi(tpe) : 0;
    const int tp = !x3 ? 4 : ((tp_env == 1 || tp_env == 2) ? tp_env : (n <= 256 ? 1 : 2));
    const unsigned grid = (unsigned)((n + tp - 1) / tp);
    if (tp == 1) hipLaunchKernelGGL((k_trunk16<true, 1>), dim3(grid), dim3(256), kLds1, stream, a, sb, ob, lg, n, n_valid, logp, v);
    else if (tp == 2) hipLaunchKernelGGL((k_trunk16<true, 2>), dim3(grid), dim3(256), kLds2, stream, a, sb, ob, lg, n, n_valid, logp, v);
    else hipLaunchKernelGGL((k_trunk16<false, 4>), dim3(grid), dim3(256), kLdsBytes, stream, a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipGetLastError());
#ifdef OTH_STAMPS
    {
        OTH_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)dbg_grid * 4 * 8);
        OTH_HIP(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double s[7] = {0, 0, 0, 0, 0, 0, 0};
        unsigned long long r0 = ~0ull, r1 = 0;
        for (size_t w = 0; w < (size_t)dbg_grid * 4; ++w) {
            for (int i = 0; i < 7; ++i) s[i] += (double)h[w * 8 + i];
            if (h[w * 8 + 5] == 0) continue;   // workgroup beyond n_valid
            if (h[w * 8 + 7] < r0) r0 = h[w * 8 + 7];
            if (h[w * 8 + 7] + h[w * 8 + 6] > r1) r1 = h[w * 8 + 7] + h[w * 8 + 6];
        }
        const double nw = (double)dbg_grid * 4;
        fprintf(stderr, "[stamps] in-kernel clock %.3f GHz (cycles / 100 MHz ticks); first start -> last end %.3f ms\n",
                s[5] / s[6] * 0.1, (double)(r1 - r0) * 1e-5);
        fprintf(stderr, "[stamps] per-wave cycles: prefetch %.0f | barrier1 %.0f | epilogue %.0f | barrier2 %.0f | conv %.0f | total %.0f\n",
                s[0] / nw, s[1] / nw, s[2] / nw, s[3] / nw, s[4] / nw, s[5] / nw);
        (void)hipFree(a.dbg);
        unsigned long long tl[128];
        OTH_HIP(hipMemcpy(tl, a.tl, sizeof(tl), hipMemcpyDeviceToHost));
        (void)hipFree(a.tl);
        if (getenv("OTH_TIMELINE") && tl[0]) {
            fprintf(stderr, "[timeline] layer: wg0 conv start,end | wg256 conv start,end (kilocycles from wg0's first conv)\n");
            for (int l = 0; l < 20; ++l)
                fprintf(stderr, "[timeline] %2d: %8.1f %8.1f | %8.1f %8.1f\n", l, (double)(long long)(tl[l * 2] - tl[0]) * 1e-3,
                        (double)(long long)(tl[l * 2 + 1] - tl[0]) * 1e-3, (double)(long long)(tl[64 + l * 2] - tl[0]) * 1e-3,
                        (double)(long long)(tl[64 + l * 2 + 1] - tl[0]) * 1e-3);
        }
    }
#endif
    return OTH_OK;
}

}  // namespace oth
