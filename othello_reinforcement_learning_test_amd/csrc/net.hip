// net.hip -- section 3 of include/othello_mi355x.h: evaluator object, weight loading (BatchNorm
// folding) and the generic fp32 kernel used for filter counts other than 128 and as an independent
// cross-check of the MFMA trunk (net_mfma.hip).
//
// Reference: /root/reference/src/model/net.py:139-205 (OthelloResNet.forward, eval mode).
#include <math.h>
#include <string.h>

#include "net.h"
#include "net_heads.h"

namespace oth {

// ------------------------------------------------------------------------------------------------
// state_dict blob -> folded host weights
// ------------------------------------------------------------------------------------------------
static const float* fold_conv_bn(FoldedConv& c, int cin, int cout, int k, const float* p) {
    c.cin = cin; c.cout = cout; c.taps = k * k;
    const float* w = p;                                   // conv.weight [cout][cin][k][k] (net.py:24)
    const float* g = p + (size_t)cout * cin * c.taps;     // bn.weight
    const float *b = g + cout, *mean = g + 2 * cout, *var = g + 3 * cout;
    c.w.resize((size_t)c.taps * cin * cout);
    c.bias.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const float scale = g[o] / sqrtf(var[o] + 1e-5f);  // nn.BatchNorm2d default eps
        c.bias[o] = b[o] - mean[o] * scale;
        for (int i = 0; i < cin; ++i)
            for (int t = 0; t < c.taps; ++t)
                c.w[((size_t)t * cin + i) * cout + o] = w[((size_t)o * cin + i) * c.taps + t] * scale;
    }
    return var + cout;
}

static int64_t state_floats(int B, int F) {
    int64_t n = 0;
    n += (int64_t)F * 3 * 9 + 4 * F;
    n += (int64_t)2 * B * ((int64_t)F * F * 9 + 4 * F);
    n += (int64_t)2 * F + 4 * 2 + 65 * 128 + 65;
    n += (int64_t)F + 4 * 1 + 256 * 64 + 256 + 256 + 1;
    return n;
}

static void parse_blob(HostNet& h, int B, int F, const float* p) {
    h.blocks = B; h.filters = F;
    p = fold_conv_bn(h.stem, 3, F, 3, p);
    h.res.assign(2 * B, FoldedConv());
    for (int i = 0; i < 2 * B; ++i) p = fold_conv_bn(h.res[i], F, F, 3, p);
    p = fold_conv_bn(h.pconv, F, 2, 1, p);
    h.pfc_w.assign(p, p + 65 * 128); p += 65 * 128;
    h.pfc_b.assign(p, p + 65); p += 65;
    p = fold_conv_bn(h.vconv, F, 1, 1, p);
    h.vfc1_w.assign(p, p + 256 * 64); p += 256 * 64;
    h.vfc1_b.assign(p, p + 256); p += 256;
    h.vfc2_w.assign(p, p + 256); p += 256;
    h.vfc2_b.assign(p, p + 1); p += 1;
}

// ------------------------------------------------------------------------------------------------
// generic fp32 kernel: one position per 256-thread block, activations [64][F] in LDS, thread owns
// one output channel and F/4 cells; weights stream from L2 as [tap][cin][cout] (coalesced over cout)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxLayers = 48;
struct GenericArgs {
    int F, n_layers;  // n_layers = 1 + 2*blocks trunk convs
    const float* w[kMaxLayers];
    const float* b[kMaxLayers];
    HeadParams heads;
};

template <int CH>  // cells processed together per thread
__device__ void conv3x3_layer(const float* __restrict__ w, const float* __restrict__ bias, int cin, int F,
                              const float* in, float* out, const float* res) {
    const int t = threadIdx.x;
    const int co = t & (F - 1);
    const int grp = t / F;        // cell group of this thread
    const int ngrp = 256 / F;     // groups; thread handles cells grp, grp+ngrp, ...
    const int per = 64 / ngrp;
    const float bv = bias[co];
    for (int c0 = 0; c0 < per; c0 += CH) {
        float acc[CH];
        int cy[CH], cx[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int cell = grp + (c0 + j) * ngrp;
            cy[j] = cell >> 3; cx[j] = cell & 7;
            acc[j] = 0.f;
        }
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            int src[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int y = cy[j] + dy, x = cx[j] + dx;
                src[j] = (y >= 0 && y < 8 && x >= 0 && x < 8) ? (y * 8 + x) * cin : -1;
            }
            const float* wt = w + (size_t)tap * cin * F + co;
            for (int ci = 0; ci < cin; ++ci) {
                const float wv = wt[(size_t)ci * F];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float av = src[j] >= 0 ? in[src[j] + ci] : 0.f;
                    acc[j] = fmaf(av, wv, acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int cell = grp + (c0 + j) * ngrp;
            float vv = acc[j] + bv;
            if (res) vv += res[cell * F + co];
            out[cell * F + co] = vv > 0.f ? vv : 0.f;  // every trunk conv output is followed by ReLU
        }
    }
}

__global__ __launch_bounds__(256) void k_net_generic(GenericArgs a, const uint64_t* __restrict__ sb,
                                                     const uint64_t* __restrict__ ob,
                                                     const uint64_t* __restrict__ lg, int64_t n,
                                                     const int32_t* __restrict__ n_valid, float* __restrict__ logp,
                                                     float* __restrict__ v) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = a.F;
    float* A = smem;
    float* B = A + 64 * F;
    float* Cb = B + 64 * F;
    float* scratch = Cb + 64 * F;  // 192 input floats, later head scratch (520 floats)
    int64_t nv = n;
    if (n_valid) {
        const int64_t k = *n_valid;
        nv = k < n ? k : n;
    }
    const int t = threadIdx.x;
    for (int64_t pos = blockIdx.x; pos < nv; pos += gridDim.x) {
        if (t < 192) {  // unpack the three input planes: in[cell][3]
            const int cell = t & 63, ch = t >> 6;
            const uint64_t bits = ch == 0 ? sb[pos] : (ch == 1 ? ob[pos] : lg[pos]);
            scratch[cell * 3 + ch] = (bits >> cell) & 1ULL ? 1.0f : 0.0f;
        }
        __syncthreads();
        conv3x3_layer<4>(a.w[0], a.b[0], 3, F, scratch, A, nullptr);  // net.py:195
        __syncthreads();
        for (int l = 1; l < a.n_layers; l += 2) {                     // net.py:198-199
            conv3x3_layer<4>(a.w[l], a.b[l], F, F, A, B, nullptr);
            __syncthreads();
            conv3x3_layer<4>(a.w[l + 1], a.b[l + 1], F, F, B, Cb, A);
            __syncthreads();
            float* tmp = A; A = Cb; Cb = tmp;
        }
        heads_forward(a.heads, F, A, F, scratch, logp + pos * 65, v + pos);
    }
}

__global__ void k_planes_to_bits(const float* __restrict__ x, uint64_t* __restrict__ sb, uint64_t* __restrict__ ob,
                                 uint64_t* __restrict__ lg, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nw) {
        const float* p = x + i * 192;
        const uint64_t a = __ballot(p[lane] > 0.5f), b = __ballot(p[64 + lane] > 0.5f), c = __ballot(p[128 + lane] > 0.5f);
        if (lane == 0) {
            sb[i] = a; ob[i] = b; lg[i] = c;
        }
    }
}

}  // namespace oth

using namespace oth;

static void net_free_device(oth_net* net) {
    if (net->d_generic) (void)hipFree(net->d_generic);
    net->d_generic = nullptr;
    mfma_free_weights(net);
}

extern "C" {

oth_net* oth_net_create(int num_blocks, int num_filters, int board_size) {
    if (board_size != 8) {
        set_error("oth_net_create: board_size %d unsupported (the reference implements 8x8 rules only)", board_size);
        return nullptr;
    }
    if (num_blocks < 1 || 1 + 2 * num_blocks > kMaxLayers) {
        set_error("oth_net_create: num_blocks %d out of range [1,%d]", num_blocks, (kMaxLayers - 1) / 2);
        return nullptr;
    }
    if (!(num_filters == 16 || num_filters == 32 || num_filters == 64 || num_filters == 128)) {
        set_error("oth_net_create: num_filters %d unsupported (16, 32, 64 or 128)", num_filters);
        return nullptr;
    }
    oth_net* n = new oth_net();
    n->device = current_device();
    n->blocks = num_blocks;
    n->filters = num_filters;
    return n;
}

void oth_net_destroy(oth_net* net) {
    if (!net) return;
    (void)bind_device(net->device);
    net_free_device(net);
    delete net;
}

int64_t oth_net_state_floats(const oth_net* net) { return net ? state_floats(net->blocks, net->filters) : 0; }

int oth_net_load_state(oth_net* net, const float* blob, int64_t n_floats, int precision) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net && blob, "oth_net_load_state: null argument");
    OTH_BIND(net->device);
    OTH_CHECK(n_floats == state_floats(net->blocks, net->filters),
              "oth_net_load_state: got %lld floats, a %dx%d network has %lld", (long long)n_floats, net->blocks,
              net->filters, (long long)state_floats(net->blocks, net->filters));
    OTH_CHECK(precision == OTH_PREC_F32 || precision == OTH_PREC_F16X3 || precision == OTH_PREC_F16,
              "oth_net_load_state: unknown precision %d", precision);
    if (precision != OTH_PREC_F32 && net->filters != 128) {
        set_error("oth_net_load_state: the MFMA kernel is built for 128 filters; use OTH_PREC_F32 for %d", net->filters);
        return OTH_E_UNSUPPORTED;
    }
    parse_blob(net->host, net->blocks, net->filters, blob);
    net_free_device(net);
    // ---- upload the folded fp32 weights (generic trunk + heads) as one allocation
    const HostNet& h = net->host;
    std::vector<float> flat;
    auto push = [&](const std::vector<float>& v) {
        size_t off = flat.size();
        flat.insert(flat.end(), v.begin(), v.end());
        while (flat.size() % 4) flat.push_back(0.f);  // keep every array 16-byte aligned
        return off;
    };
    net->conv_w_off.clear();
    net->conv_b_off.clear();
    net->conv_w_off.push_back(push(h.stem.w));
    net->conv_b_off.push_back(push(h.stem.bias));
    for (const auto& c : h.res) {
        net->conv_w_off.push_back(push(c.w));
        net->conv_b_off.push_back(push(c.bias));
    }
    const size_t o_pw = push(h.pconv.w), o_pb = push(h.pconv.bias), o_vw = push(h.vconv.w), o_vb = push(h.vconv.bias);
    const size_t o_pfw = push(h.pfc_w), o_pfb = push(h.pfc_b), o_v1w = push(h.vfc1_w), o_v1b = push(h.vfc1_b);
    const size_t o_v2w = push(h.vfc2_w), o_v2b = push(h.vfc2_b);
    OTH_HIP(hipMalloc(&net->d_generic, flat.size() * sizeof(float)));
    OTH_HIP(hipMemcpy(net->d_generic, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* d = net->d_generic;
    net->heads = HeadParams{d + o_pw, d + o_pb, d + o_vw, d + o_vb, d + o_pfw, d + o_pfb, d + o_v1w, d + o_v1b, d + o_v2w, d + o_v2b};
    if (precision != OTH_PREC_F32) {
        int r = mfma_pack_weights(net, precision);
        if (r != OTH_OK) return r;
    }
    net->precision = precision;
    return OTH_OK;
}

int oth_net_forward_bits(oth_net* net, const uint64_t* sb, const uint64_t* ob, const uint64_t* lg, int64_t n,
                         const int32_t* n_valid, float* logp, float* v, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net && net->precision >= 0, "oth_net_forward: no weights loaded (call oth_net_load_state)");
    OTH_CHECK(n >= 0 && (n == 0 || (sb && ob && lg && logp && v)), "oth_net_forward: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND(net->device);
    if (net->precision != OTH_PREC_F32) return mfma_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
    GenericArgs a;
    memset(&a, 0, sizeof(a));
    a.F = net->filters;
    a.n_layers = 1 + 2 * net->blocks;
    for (int l = 0; l < a.n_layers; ++l) {
        a.w[l] = net->d_generic + net->conv_w_off[l];
        a.b[l] = net->d_generic + net->conv_b_off[l];
    }
    a.heads = net->heads;
    const size_t lds = (size_t)(3 * 64 * net->filters + 640) * sizeof(float);
    static bool attr_set_dev[64] = {};  // per device: the attribute belongs to the (function, device) pair
    bool& attr_set = attr_set_dev[net->device & 63];
    if (!attr_set) {
        OTH_HIP(hipFuncSetAttribute((const void*)k_net_generic, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int grid = (int)(n < 2048 ? n : 2048);
    hipLaunchKernelGGL(k_net_generic, dim3(grid), dim3(256), lds, as_stream(stream), a, sb, ob, lg, n, n_valid, logp, v);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

int oth_net_forward_planes(oth_net* net, const float* x, int64_t n, float* logp, float* v, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net && x && logp && v && n >= 0, "oth_net_forward_planes: bad arguments");
    if (n == 0) return OTH_OK;
    OTH_BIND(net->device);
    uint64_t* bits = nullptr;
    OTH_HIP(hipMallocAsync((void**)&bits, sizeof(uint64_t) * 3 * n, as_stream(stream)));
    int64_t g = (n + 3) / 4;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_planes_to_bits, dim3((int)g), dim3(256), 0, as_stream(stream), x, bits, bits + n, bits + 2 * n, n);
    int r = oth_net_forward_bits(net, bits, bits + n, bits + 2 * n, n, nullptr, logp, v, stream);
    OTH_HIP(hipFreeAsync(bits, as_stream(stream)));
    return r;
}

}  // extern "C"
