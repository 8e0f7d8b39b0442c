// net.h -- evaluator object shared by net.hip (API, weight folding), net_f32.hip (exact-fp32 MFMA trunk, every
// width and board size) and net_mfma.hip (the fp16-split MFMA trunk for 128 filters).
#pragma once
#include <vector>

#include "common.h"

namespace oth {

// One conv+BN pair folded for inference (eval-mode BN, reference L19):
//   y = conv(x, w) * scale + shift,  scale = gamma / sqrt(var + eps),  shift = beta - mean * scale
// stored as w'[tap][cin][cout] = w * scale[cout] and bias[cout] = shift.
struct FoldedConv {
    int cin = 0, cout = 0, taps = 0;
    std::vector<float> w;     // [taps][cin][cout]
    std::vector<float> bias;  // [cout]
};

constexpr int kMaxTrunkLayers = 48;  // 1 + 2*blocks

struct HostNet {
    int blocks = 0, filters = 0, board = 8;
    FoldedConv stem;               // 3 -> F, 3x3            (net.py:168)
    std::vector<FoldedConv> res;   // 2*blocks, F -> F, 3x3  (net.py:171-173)
    FoldedConv pconv, vconv;       // 1x1 heads              (net.py:76, 111)
    std::vector<float> pfc_w, pfc_b;    // [cells+1][2*cells], [cells+1]   (net.py:81; 8x8: [65][128])
    std::vector<float> vfc1_w, vfc1_b;  // [256][cells], [256]             (net.py:116)
    std::vector<float> vfc2_w, vfc2_b;  // [256], [1]        (net.py:117)
};

// device-side parameter block of the heads (fp32, shared by both trunk kernels)
struct HeadParams {
    const float* pconv_w;  // [F][2]
    const float* pconv_b;  // [2]
    const float* vconv_w;  // [F]
    const float* vconv_b;  // [1]
    const float* pfc_w;    // [65][128]
    const float* pfc_b;    // [65]
    const float* vfc1_w;   // [256][64]
    const float* vfc1_b;   // [256]
    const float* vfc2_w;   // [256]
    const float* vfc2_b;   // [1]
};

struct MfmaWeights;  // net_mfma.hip
struct F32Weights;   // net_f32.hip
struct H3Weights;    // net_h3.hip
struct WinoWeights;  // net_wino.hip
struct Wino6Weights; // net_wino6.hip

}  // namespace oth

struct oth_net {
    int device = 0;      // HIP device the weights live on (the creating thread's current device)
    int blocks = 0, filters = 0, board = 8;
    int precision = -1;  // OTH_PREC_*; -1 = no weights loaded
    oth::HostNet host;
    float* d_heads = nullptr;  // one allocation: the fp32 head parameters (shared by both trunk kernels)
    int* d_sat = nullptr;      // device flag: an fp16-split trunk kernel clamped an activation (oth_net_saturated)
    // Activations of the fp16-split trunks are carried PRE-SCALED by act_scale (a power of two, 16 by default: it keeps the
    // lo parts of the operand split clear of the f16 subnormals) and clamped at the f16 range of that scaled value; the
    // reference's fp32 forward has no clamp.  The scale enters a launch through the stem's input planes, the biases
    // (dev = host x act_scale: the arrays registered below) and the heads' un-scaling, so it is a RUN-TIME value:
    // oth_net_set_act_scale halves it when a launch saturated (range 1875 -> 30 000 at scale 1 in the Winograd trunks).
    float act_scale = 16.0f;
    struct ScaledBias { float* dev; std::vector<float> host; };
    std::vector<ScaledBias> scaled_bias;
    oth::HeadParams heads{};
    // exact-fp32 MFMA path (any filter count, 8x8 and 6x6): net_f32.hip
    oth::F32Weights* f32 = nullptr;
    // fp16-split MFMA path (128 filters, 8x8): net_mfma.hip
    oth::MfmaWeights* mfma = nullptr;
    // fp16-split MFMA path, one wave per position (32 / 64 filters, 8x8 and 6x6): net_h3.hip
    oth::H3Weights* h3 = nullptr;
    // 1-D Winograd F(2,3) build of the 128-filter 8x8 trunk (fp16-split arithmetic): net_wino.hip
    oth::WinoWeights* wino = nullptr;
    // 1-D Winograd F(2,3) build of the 64-filter 6x6 trunk (BASELINE configs[4]): net_wino6.hip
    oth::Wino6Weights* wino6 = nullptr;
};

// The in-place MFMAs of net_mfma.hip / net_h3.hip are inline asm: hipcc inserts no wait states between a VALU write of a
// VGPR and an asm MFMA reading it (2 needed on gfx90a+).  Accumulators are zeroed by VALU moves which the compiler likes to
// sink next to their first use; OTH_PIN_ACC keeps the move of one accumulator in front of this point and
// OTH_PIN_ACC_END supplies the wait states once for all of them.  tools/check_mfma_hazards.py checks the built objects.
#define OTH_PIN_ACC(x) asm volatile("" : "+v"(x))
#define OTH_PIN_ACC_END()                  \
    do {                                   \
        asm volatile("s_nop 1" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

namespace oth {
// upload host x net->act_scale to dev and keep both for oth_net_set_act_scale (dev stays owned by the caller's weight struct)
int register_scaled_bias(oth_net* net, float* dev, std::vector<float> host);   // net.hip
int mfma_pack_weights(oth_net* net, int precision);  // net_mfma.hip
void mfma_free_weights(oth_net* net);
int mfma_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
                 const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
int h3_pack_weights(oth_net* net);  // net_h3.hip
void h3_free_weights(oth_net* net);
int h3_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
               const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
int wino6_pack_weights(oth_net* net);  // net_wino6.hip
void wino6_free_weights(oth_net* net);
int wino6_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
                  const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
int wino_pack_weights(oth_net* net);  // net_wino.hip
void wino_free_weights(oth_net* net);
int wino_positions_per_workgroup(int64_t n);   // 1 or 2: which k_trunk_w build a launch of n positions runs
int wino_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
                 const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
int f32_pack_weights(oth_net* net);  // net_f32.hip
void f32_free_weights(oth_net* net);
int f32_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
                const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
}  // namespace oth
