// net_mfma.hip -- OthelloResNet forward for 128 filters as ONE fused gfx950 kernel.
//
// Reference: /root/reference/src/model/net.py:182-205 (eval mode; BatchNorm folded at load time).
//
// Design (MI355X-first; see DESIGN.md "K3"):
//   * one 256-thread workgroup (4 waves, one per SIMD, whole 512-register file each) evaluates a
//     tile of 4 positions = 256 GEMM rows through the stem, all residual blocks and both heads;
//     the activations NEVER leave the CU: they live in LDS as [cell][hi 128 x f16 | lo 128 x f16]
//     (512 B per cell, 128 KiB per tile) and every 3x3 tap reads them in place (implicit GEMM, no
//     im2col copy); out-of-board taps read a zeroed cell.
//   * wave w owns output channels [32w, 32w+32): its B operand (weights) streams from L2 straight
//     into registers in MFMA fragment order (host-packed, 1 KiB per wave-load, every byte loaded
//     exactly once per workgroup and layer); accumulators (8 tiles of 32x32) and the fp32 residual
//     stay in registers, so there is no barrier inside a layer -- two per layer around the
//     in-place activation rewrite.
//   * arithmetic: v_mfma_f32_32x32x16_f16 with both operands split a = a_hi + a_lo (f16 pairs,
//     ~22 significant bits), three products a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in fp32:
//     fp32-equivalent results (the 1e-4 parity tolerance leaves no room for a single f16 pass in
//     general).  OTH_PREC_F16 runs the same kernel with the hi parts only.
//   * power-of-two scaling of activations (2^4) and of each layer's weights keeps the lo parts in
//     the normal f16 range; it is undone exactly on the fp32 accumulator.
//   * LDS bank conflicts: the 16-byte chunk index of a cell is XORed with (x&3)|((y&3)<<2); the 16
//     lanes of every ds_read_b128 group then hit 16 different slots for all nine taps.
#include <math.h>
#include <string.h>

#include <vector>

#include "net.h"
#include "net_heads.h"

namespace oth {

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int kTilePos = 4;                    // positions per workgroup
constexpr int kCellBytes = 512;                // 256 B hi + 256 B lo
constexpr int kActBytes = 256 * kCellBytes;    // 131072
constexpr int kZeroOff = kActBytes;            // zero cell (512 B)
constexpr int kScratchOff = kActBytes + 512;   // stem im2col (16 KiB) / head scratch
constexpr int kLdsBytes = kScratchOff + 16384;
constexpr float kActScale = 16.0f;             // 2^4

struct MfmaWeights {
    int blocks = 0;
    uint4* d_w = nullptr;      // fragments: [layer][step 0..71][wave 0..3][plane hi,lo][64 lanes] x 16 B
    uint4* d_stem = nullptr;   // [step 0..1][wave][plane][64 lanes]
    float* d_bias = nullptr;   // [1 + 2*blocks][128]
    float* d_inv = nullptr;    // [1 + 2*blocks] 1 / (weight_scale * act_scale)
};

struct MfmaArgs {
    const uint4* w;
    const uint4* stem;
    const float* bias;
    const float* inv;
    int n_res_layers;  // 2 * blocks
    HeadParams heads;
};

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <bool X3>
__global__ __launch_bounds__(256, 1) void k_trunk(MfmaArgs a, const uint64_t* __restrict__ sb,
                                                  const uint64_t* __restrict__ ob,
                                                  const uint64_t* __restrict__ lgl, int64_t n,
                                                  const int32_t* __restrict__ n_valid, float* __restrict__ logp,
                                                  float* __restrict__ vout) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int64_t pos0 = (int64_t)blockIdx.x * kTilePos;
    if (pos0 >= nv) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, r = lane & 31;

    // ---------------- stem input: im2col of the three bit planes, [256 cells][32 k] f16, k = tap*3 + plane
    {
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
            vals[tap * 3 + 0] = (ok && ((b0 >> s) & 1ULL)) ? (_Float16)kActScale : (_Float16)0.0f;
            vals[tap * 3 + 1] = (ok && ((b1 >> s) & 1ULL)) ? (_Float16)kActScale : (_Float16)0.0f;
            vals[tap * 3 + 2] = (ok && ((b2 >> s) & 1ULL)) ? (_Float16)kActScale : (_Float16)0.0f;
        }
        half8* dst = (half8*)(lds + kScratchOff + tid * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8 t;
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = vals[q * 8 + j];
            dst[q] = t;
        }
        if (tid < 32) ((uint4*)(lds + kZeroOff))[tid] = make_uint4(0, 0, 0, 0);  // the zero cell
    }
    __syncthreads();

    f32x16 acc[8], res[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---------------- stem conv as a K=32 GEMM (net.py:195)
    {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint4* wp = a.stem + ((size_t)(kk * 4 + wave) * 2) * 64 + lane;
            const uint4 bh4 = wp[0], bl4 = wp[64];
            const half8 bh = __builtin_bit_cast(half8, bh4), bl = __builtin_bit_cast(half8, bl4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const half8 ah = *(const half8*)(lds + kScratchOff + (t * 32 + r) * 64 + (kk * 2 + h) * 16);
                if (X3) acc[t] = mfma16(ah, bl, acc[t]);
                acc[t] = mfma16(ah, bh, acc[t]);
            }
        }
    }

    // write-side constants of this lane: output channel n, its 16-byte chunk and byte offset in the chunk
    const int n_out = wave * 32 + r;
    const int chunk_n = n_out >> 3;
    const uint32_t wr_lane = (uint32_t)(4 * h) * kCellBytes + (uint32_t)(n_out & 7) * 2;

    const int n_layers = 1 + a.n_res_layers;
    for (int layer = 0; layer < n_layers; ++layer) {
        // ---------------- epilogue of conv `layer` (0 = stem): scale back, bias, skip, ReLU, re-split
        const float bias_n = a.bias[layer * 128 + n_out];
        const float inv = a.inv[layer];
        const bool add_res = layer > 0 && (layer & 1) == 0;   // second conv of a block (net.py:58)
        const bool set_res = layer == 0 || add_res;
        __syncthreads();  // every wave has finished reading the previous activations
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = fmaf(acc[t][i], inv, bias_n);
                if (add_res) v += res[t][i];
                v = v > 0.f ? v : 0.f;
                if (set_res) res[t][i] = v;
                acc[t][i] = 0.f;
                const float vs = fminf(v * kActScale, 60000.0f);
                const _Float16 hi = (_Float16)vs;
                const uint32_t cell_off = (uint32_t)(t * 32 + (i & 3) + 8 * (i >> 2)) * kCellBytes;
                const uint32_t addr = cell_off + wr_lane + (uint32_t)((chunk_n ^ i) << 4);
                *(_Float16*)(lds + addr) = hi;
                if (X3) *(_Float16*)(lds + addr + 256) = (_Float16)(vs - (float)hi);
            }
        }
        __syncthreads();
        if (layer == n_layers - 1) break;

        // ---------------- conv `layer+1`: 9 taps x 8 k-steps of 16 input channels
        const uint4* wl = a.w + ((size_t)layer * 72 * 4 + wave) * 128 + lane;  // + step*4*128
        uint4 bh4 = wl[0], bl4 = wl[64];
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            // per-tile base address of this lane's source cell (or the zero cell)
            const int yo = (r >> 3) + dy, xs = (r & 7) + dx;   // yo relative to the tile's first row
            const bool xok = xs >= 0 && xs < 8;
            const uint32_t hk = (uint32_t)(h ^ ((xs & 3) | ((yo & 3) << 2)));
            uint32_t A[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int ys = (t & 1) * 4 + yo;
                const bool ok = xok && ys >= 0 && ys < 8;
                A[t] = ok ? (uint32_t)((t >> 1) * 64 + ys * 8 + xs) * kCellBytes : (uint32_t)kZeroOff;
            }
#pragma unroll 2
            for (int kk = 0; kk < 8; ++kk) {
                const half8 bh = __builtin_bit_cast(half8, bh4), bl = __builtin_bit_cast(half8, bl4);
                // prefetch the next step's fragments (the last step of a layer re-reads its own: harmless)
                const int step = tap * 8 + kk;
                const int nstep = step + 1 < 72 ? step + 1 : step;
                bh4 = wl[(size_t)nstep * 512];
                bl4 = wl[(size_t)nstep * 512 + 64];
                const uint32_t off = (((uint32_t)(kk << 1)) ^ hk) << 4;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const char* p = lds + (A[t] | off);
                    const half8 ah = *(const half8*)p;
                    if (X3) {
                        const half8 al = *(const half8*)(p + 256);
                        acc[t] = mfma16(al, bh, acc[t]);
                        acc[t] = mfma16(ah, bl, acc[t]);
                    }
                    acc[t] = mfma16(ah, bh, acc[t]);
                }
            }
        }
    }

    // ---------------- heads (fp32 VALU): final activations (in `res`) -> LDS [256 cells][128] f32
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cell = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            *(float*)(lds + (size_t)cell * 512 + n_out * 4) = res[t][i];
        }
    __syncthreads();
    for (int p = 0; p < kTilePos; ++p) {
        if (pos0 + p >= nv) break;  // uniform across the block
        heads_forward(a.heads, 128, (const float*)(lds + (size_t)p * 64 * 512), 128, (float*)(lds + kScratchOff),
                      logp + (pos0 + p) * 65, vout + pos0 + p);
    }
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

// B fragment of v_mfma_f32_32x32x16_f16: lane l holds B[k = 8*(l>>5) + j][col = l&31], j = 0..7
static float pack_conv(const FoldedConv& c, int k_total, std::vector<uint16_t>& out, size_t base, int steps,
                       const std::vector<int>& k_map /* gemm k -> index into [tap*cin + ci], -1 = zero */) {
    float mx = 0.f;
    for (float x : c.w) mx = fmaxf(mx, fabsf(x));
    int e = 0;
    if (mx > 0.f) e = (int)floorf(log2f(16384.0f / mx));  // largest |w| lands in [8192, 16384]
    if (e > 24) e = 24;
    if (e < -24) e = -24;
    const float scale = ldexpf(1.0f, e);
    (void)k_total;
    for (int s = 0; s < steps; ++s)
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int k = s * 16 + 8 * (l >> 5) + j, col = w * 32 + (l & 31);
                    const int src = k_map[k];
                    const float v = src < 0 ? 0.f : c.w[(size_t)src * c.cout + col] * scale;
                    uint16_t hi, lo;
                    split_f16(v, hi, lo);
                    const size_t frag = base + ((size_t)(s * 4 + w) * 2) * 64 * 8;  // in halfs
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
    {   // stem: gemm k = tap*3 + plane for k < 27 == index into [tap][cin=3]
        std::vector<int> km(32, -1);
        for (int k = 0; k < 27; ++k) km[k] = k;
        const float sc = pack_conv(hn.stem, 32, stem, 0, 2, km);
        inv[0] = 1.0f / (sc * kActScale);
        memcpy(&bias[0], hn.stem.bias.data(), 128 * sizeof(float));
    }
    std::vector<int> km(1152);
    for (int k = 0; k < 1152; ++k) km[k] = k;  // k = tap*128 + ci, steps ordered (tap, kk)
    for (int l = 0; l < L; ++l) {
        const float sc = pack_conv(hn.res[l], 1152, w, (size_t)l * layer_halfs, 72, km);
        inv[l + 1] = 1.0f / (sc * kActScale);
        memcpy(&bias[(size_t)(l + 1) * 128], hn.res[l].bias.data(), 128 * sizeof(float));
    }
    OTH_HIP(hipMalloc(&mw->d_w, w.size() * 2));
    OTH_HIP(hipMalloc(&mw->d_stem, stem.size() * 2));
    OTH_HIP(hipMalloc(&mw->d_bias, bias.size() * 4));
    OTH_HIP(hipMalloc(&mw->d_inv, inv.size() * 4));
    OTH_HIP(hipMemcpy(mw->d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(mw->d_stem, stem.data(), stem.size() * 2, hipMemcpyHostToDevice));
    OTH_HIP(hipMemcpy(mw->d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
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
    static bool attr_set = false;
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        OTH_HIP(hipFuncSetAttribute((const void*)k_trunk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        attr_set = true;
    }
    const unsigned grid = (unsigned)((n + kTilePos - 1) / kTilePos);
    if (net->precision == OTH_PREC_F16X3)
        hipLaunchKernelGGL(k_trunk<true>, dim3(grid), dim3(256), kLdsBytes, stream, a, sb, ob, lg, n, n_valid, logp, v);
    else
        hipLaunchKernelGGL(k_trunk<false>, dim3(grid), dim3(256), kLdsBytes, stream, a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

}  // namespace oth
