// net.h -- evaluator object shared by net.hip (generic fp32 kernel + API) and net_mfma.hip
// (the F=128 MFMA trunk kernel).
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

struct HostNet {
    int blocks = 0, filters = 0;
    FoldedConv stem;               // 3 -> F, 3x3            (net.py:168)
    std::vector<FoldedConv> res;   // 2*blocks, F -> F, 3x3  (net.py:171-173)
    FoldedConv pconv, vconv;       // 1x1 heads              (net.py:76, 111)
    std::vector<float> pfc_w, pfc_b;    // [65][128], [65]   (net.py:81)
    std::vector<float> vfc1_w, vfc1_b;  // [256][64], [256]  (net.py:116)
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

}  // namespace oth

struct oth_net {
    int device = 0;      // HIP device the weights live on (the creating thread's current device)
    int blocks = 0, filters = 0;
    int precision = -1;  // OTH_PREC_*; -1 = no weights loaded
    oth::HostNet host;
    // generic fp32 path
    float* d_generic = nullptr;  // one allocation: all folded convs + heads
    std::vector<size_t> conv_w_off, conv_b_off;  // per conv layer (stem, res...), float offsets
    oth::HeadParams heads{};
    // MFMA path (filters == 128)
    oth::MfmaWeights* mfma = nullptr;
};

namespace oth {
int mfma_pack_weights(oth_net* net, int precision);  // net_mfma.hip
void mfma_free_weights(oth_net* net);
int mfma_forward(oth_net* net, const uint64_t* self_b, const uint64_t* opp_b, const uint64_t* legal, int64_t n,
                 const int32_t* n_valid, float* logp, float* v, hipStream_t stream);
}  // namespace oth
