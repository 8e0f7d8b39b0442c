// net.hip -- section 3 of include/othello_mi355x.h: evaluator object, weight loading (BatchNorm folding) and
// dispatch to the two trunk kernels: net_f32.hip (exact fp32 on v_mfma_f32_16x16x4_f32: every filter count,
// 8x8 and 6x6; also the independent cross-check of the split kernel) and net_mfma.hip (fp16 hi/lo split on
// v_mfma_f32_16x16x32_f16: 128 filters on 8x8, the benchmarked configuration).
//
// Reference: /root/reference/src/model/net.py:139-205 (OthelloResNet.forward, eval mode).
#include <math.h>
#include <string.h>

#include <stdlib.h>

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

static int64_t state_floats(int B, int F, int board) {
    const int64_t cells = (int64_t)board * board;
    int64_t n = 0;
    n += (int64_t)F * 3 * 9 + 4 * F;
    n += (int64_t)2 * B * ((int64_t)F * F * 9 + 4 * F);
    n += (int64_t)2 * F + 4 * 2 + (cells + 1) * 2 * cells + (cells + 1);   // net.py:76-81
    n += (int64_t)F + 4 * 1 + 256 * cells + 256 + 256 + 1;                 // net.py:111-117
    return n;
}

static void parse_blob(HostNet& h, int B, int F, int board, const float* p) {
    h.blocks = B; h.filters = F; h.board = board;
    const int cells = board * board;
    p = fold_conv_bn(h.stem, 3, F, 3, p);
    h.res.assign(2 * B, FoldedConv());
    for (int i = 0; i < 2 * B; ++i) p = fold_conv_bn(h.res[i], F, F, 3, p);
    p = fold_conv_bn(h.pconv, F, 2, 1, p);
    h.pfc_w.assign(p, p + (size_t)(cells + 1) * 2 * cells); p += (size_t)(cells + 1) * 2 * cells;
    h.pfc_b.assign(p, p + cells + 1); p += cells + 1;
    p = fold_conv_bn(h.vconv, F, 1, 1, p);
    h.vfc1_w.assign(p, p + (size_t)256 * cells); p += (size_t)256 * cells;
    h.vfc1_b.assign(p, p + 256); p += 256;
    h.vfc2_w.assign(p, p + 256); p += 256;
    h.vfc2_b.assign(p, p + 1); p += 1;
}

__global__ void k_planes_to_bits(const float* __restrict__ x, uint64_t* __restrict__ sb, uint64_t* __restrict__ ob,
                                 uint64_t* __restrict__ lg, int64_t n, int cells) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool in = lane < cells;
    const int l = in ? lane : 0;
    for (int64_t i = wave; i < n; i += nw) {
        const float* p = x + i * 3 * cells;
        const uint64_t a = __ballot(in && p[l] > 0.5f), b = __ballot(in && p[cells + l] > 0.5f),
                       c = __ballot(in && p[2 * cells + l] > 0.5f);
        if (lane == 0) {
            sb[i] = a; ob[i] = b; lg[i] = c;
        }
    }
}

int register_scaled_bias(oth_net* net, float* dev, std::vector<float> host) {
    std::vector<float> up(host.size());
    for (size_t i = 0; i < host.size(); ++i) up[i] = host[i] * net->act_scale;
    OTH_HIP(hipMemcpy(dev, up.data(), up.size() * sizeof(float), hipMemcpyHostToDevice));
    net->scaled_bias.push_back({dev, std::move(host)});
    return OTH_OK;
}

}  // namespace oth

using namespace oth;

static void net_free_device(oth_net* net) {
    net->scaled_bias.clear();   // (the device arrays belong to the weight structs freed below)
    if (net->d_heads) (void)hipFree(net->d_heads);
    net->d_heads = nullptr;
    if (net->d_sat) (void)hipFree(net->d_sat);
    net->d_sat = nullptr;
    f32_free_weights(net);
    h3_free_weights(net);
    mfma_free_weights(net);
    wino_free_weights(net);
    wino6_free_weights(net);
}

extern "C" {

oth_net* oth_net_create(int num_blocks, int num_filters, int board_size) {
    if (board_size != 8 && board_size != 6) {  // net.py:157-180 is size-parametric; the reference's configs use 8 and 6
        set_error("oth_net_create: board_size %d unsupported (8 or 6)", board_size);
        return nullptr;
    }
    if (num_blocks < 1 || 1 + 2 * num_blocks > kMaxTrunkLayers) {
        set_error("oth_net_create: num_blocks %d out of range [1,%d]", num_blocks, (kMaxTrunkLayers - 1) / 2);
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
    n->board = board_size;
    return n;
}

void oth_net_destroy(oth_net* net) {
    if (!net) return;
    (void)bind_device(net->device);
    net_free_device(net);
    delete net;
}

int oth_net_saturated(oth_net* net, int32_t* flag, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net && flag, "oth_net_saturated: null argument");
    *flag = 0;
    if (!net->d_sat) return OTH_OK;   // no weights loaded yet
    OTH_BIND(net->device);
    hipStream_t s = as_stream(stream);
    int v = 0;
    OTH_HIP(hipMemcpyAsync(&v, net->d_sat, sizeof(int), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    if (v) OTH_HIP(hipMemsetAsync(net->d_sat, 0, sizeof(int), s));
    *flag = v;
    return OTH_OK;
}

int oth_net_kernel_info(const oth_net* net, int64_t n, char* name, int32_t name_cap, double* issued_per_flop,
                        double* clamp) {
    OTH_CHECK(net && net->precision >= 0, "oth_net_kernel_info: no weights loaded (call oth_net_load_state)");
    // the same order of tests as oth_net_forward_bits below
    const char* k;
    double issued, cl = 0.0;
    if (net->precision == OTH_PREC_F32) {
        k = "k_trunk_f32 (fused ResNet forward, exact fp32 on v_mfma_f32_16x16x4_f32, one wave per position group)";
        issued = 1.0;
    } else if (net->wino6) {
        k = "k_trunk_w6 (fused ResNet forward, 6x6, 1-D Winograd F(2,3) residual convolutions, eight positions per workgroup)";
        issued = 3.0 * 2.0 / 3.0;   // three split products, 4 multiplies per 2 outputs instead of 6, every N-tile full
        cl = 30000.0 / net->act_scale;   // 1875 at the default scale 16
    } else if (net->h3) {
        k = "k_trunk_h3 (fused ResNet forward, direct 3x3, one wave per position group)";
        // three split products; N-tiles of 16 cells over the wave's positions (P = 2 on 6x6 with 64 filters, 4 with 32, else 1)
        const int cells = net->board * net->board;
        const int P = net->board == 6 ? (net->filters == 64 ? 2 : net->filters == 32 ? 4 : 1) : 1;
        issued = 3.0 * (double)((P * cells + 15) / 16 * 16) / (double)(P * cells);
        cl = 60000.0 / net->act_scale;   // 3750 at the default scale 16
    } else if (net->wino) {
        k = wino_positions_per_workgroup(n) == 2 ? "k_trunk_w<2> (fused ResNet forward, 1-D Winograd F(2,3) residual convolutions, two positions per workgroup)"
                    : "k_trunk_w<1> (fused ResNet forward, 1-D Winograd F(2,3) residual convolutions, one position per workgroup)";
        issued = 3.0 * 2.0 / 3.0;
        cl = 30000.0 / net->act_scale;   // 1875 at the default scale 16
    } else {
        const bool x3 = net->precision != OTH_PREC_F16;
        k = x3 ? "k_trunk16 (fused ResNet forward, direct 3x3, fp16 hi/lo split)" : "k_trunk16 (fused ResNet forward, direct 3x3, single fp16 pass)";
        issued = (x3 ? 3.0 : 1.0) * 11.0 / 12.0;   // the all-padding row tiles of the dy = -1 / +1 taps are skipped
        cl = 60000.0 / net->act_scale;   // 3750 at the default scale 16
    }
    if (name && name_cap > 0) {
        strncpy(name, k, (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (issued_per_flop) *issued_per_flop = issued;
    if (clamp) *clamp = cl;
    return OTH_OK;
}

int oth_net_get_act_scale(const oth_net* net, float* scale) {
    OTH_CHECK(net && scale, "oth_net_get_act_scale: null argument");
    *scale = net->act_scale;
    return OTH_OK;
}

int oth_net_set_act_scale(oth_net* net, float scale) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net, "oth_net_set_act_scale: null network");
    OTH_CHECK(scale == 1.f || scale == 2.f || scale == 4.f || scale == 8.f || scale == 16.f,
              "oth_net_set_act_scale: the scale must be 1, 2, 4, 8 or 16 (got %g)", (double)scale);
    OTH_BIND(net->device);
    if (scale == net->act_scale) return OTH_OK;
    // the biases of the fp16-split trunks are stored x act_scale: no launch on this network may be in flight while they
    // are rewritten (callers change the scale between calls; this makes sure)
    OTH_HIP(hipDeviceSynchronize());
    net->act_scale = scale;
    std::vector<float> up;
    for (auto& sbias : net->scaled_bias) {
        up.resize(sbias.host.size());
        for (size_t i = 0; i < up.size(); ++i) up[i] = sbias.host[i] * scale;
        OTH_HIP(hipMemcpy(sbias.dev, up.data(), up.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return OTH_OK;
}

int64_t oth_net_state_floats(const oth_net* net) { return net ? state_floats(net->blocks, net->filters, net->board) : 0; }
int oth_net_policy_size(const oth_net* net) { return net ? net->board * net->board + 1 : 0; }

int oth_net_load_state(oth_net* net, const float* blob, int64_t n_floats, int precision) {
    OTH_NEED_DEVICE();
    OTH_CHECK(net && blob, "oth_net_load_state: null argument");
    OTH_BIND(net->device);
    OTH_CHECK(n_floats == state_floats(net->blocks, net->filters, net->board),
              "oth_net_load_state: got %lld floats, a %dx%d network on a %dx%d board has %lld", (long long)n_floats,
              net->blocks, net->filters, net->board, net->board,
              (long long)state_floats(net->blocks, net->filters, net->board));
    OTH_CHECK(precision == OTH_PREC_F32 || precision == OTH_PREC_F16X3 || precision == OTH_PREC_F16 ||
                  precision == OTH_PREC_F16X3_DIRECT,
              "oth_net_load_state: unknown precision %d", precision);
    // fp16-split kernels: 128 filters on 8x8 (k_trunk16: f16x3 and the single-pass f16), 32 / 64 filters on either
    // board (k_trunk_h3: f16x3 only); everything else runs the exact-fp32 MFMA kernel
    const bool wide = net->filters == 128 && net->board == 8;
    const bool narrow = net->filters == 32 || net->filters == 64 || (net->filters == 128 && net->board == 6);
    if (precision != OTH_PREC_F32 && !(wide || (narrow && precision == OTH_PREC_F16X3))) {
        set_error("oth_net_load_state: no fp16-split kernel for %d filters on a %dx%d board at precision %d; use "
                  "OTH_PREC_F32 (exact fp32 MFMA)", net->filters, net->board, net->board, precision);
        return OTH_E_UNSUPPORTED;
    }
    parse_blob(net->host, net->blocks, net->filters, net->board, blob);
    net_free_device(net);
    // ---- upload the fp32 head parameters as one allocation
    const HostNet& h = net->host;
    std::vector<float> flat;
    auto push = [&](const std::vector<float>& v) {
        size_t off = flat.size();
        flat.insert(flat.end(), v.begin(), v.end());
        while (flat.size() % 4) flat.push_back(0.f);  // keep every array 16-byte aligned
        return off;
    };
    const size_t o_pw = push(h.pconv.w), o_pb = push(h.pconv.bias), o_vw = push(h.vconv.w), o_vb = push(h.vconv.bias);
    const size_t o_pfw = push(h.pfc_w), o_pfb = push(h.pfc_b), o_v1w = push(h.vfc1_w), o_v1b = push(h.vfc1_b);
    const size_t o_v2w = push(h.vfc2_w), o_v2b = push(h.vfc2_b);
    OTH_HIP(hipMalloc(&net->d_sat, sizeof(int)));
    OTH_HIP(hipMemset(net->d_sat, 0, sizeof(int)));
    OTH_HIP(hipMalloc(&net->d_heads, flat.size() * sizeof(float)));
    OTH_HIP(hipMemcpy(net->d_heads, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* d = net->d_heads;
    net->heads = HeadParams{d + o_pw, d + o_pb, d + o_vw, d + o_vb, d + o_pfw, d + o_pfb, d + o_v1w, d + o_v1b, d + o_v2w, d + o_v2b};
    int r = precision == OTH_PREC_F32 ? f32_pack_weights(net)
                                      : (wide ? mfma_pack_weights(net, precision) : h3_pack_weights(net));
    if (r != OTH_OK) return r;
    // A 128-filter 8x8 network in f16x3 runs the 1-D Winograd F(2,3) trunk (net_wino.hip: 1.375x fewer MFMAs, +9 % games/s
    // on the bench; its one-position build serves the launches of <= 256 positions, so a position's outputs do not depend
    // on the launch size); OTH_WINO=0 keeps the direct kernel (k_trunk16) -- the A/B switch.
    const char* wino_env = getenv("OTH_WINO");
    if (precision == OTH_PREC_F16X3 && wide && net->board == 8 && !(wino_env && atoi(wino_env) == 0)) {
        r = wino_pack_weights(net);
        if (r != OTH_OK) return r;
    }
    // A 64-filter 6x6 network (BASELINE configs[4]) in f16x3 runs the 6x6 Winograd trunk (net_wino6.hip: eight positions per
    // workgroup, 1.5x fewer MFMAs and no padded cells); OTH_WINO6=0 keeps k_trunk_h3 -- the A/B switch.
    const char* wino6_env = getenv("OTH_WINO6");
    if (precision == OTH_PREC_F16X3 && net->filters == 64 && net->board == 6 && !(wino6_env && atoi(wino6_env) == 0)) {
        r = wino6_pack_weights(net);
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
    if (net->precision == OTH_PREC_F32) return f32_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
    if (net->wino6) return wino6_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
    if (net->h3) return h3_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
    if (net->wino) return wino_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
    return mfma_forward(net, sb, ob, lg, n, n_valid, logp, v, as_stream(stream));
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
    hipLaunchKernelGGL(k_planes_to_bits, dim3((int)g), dim3(256), 0, as_stream(stream), x, bits, bits + n, bits + 2 * n, n,
                       net->board * net->board);
    int r = oth_net_forward_bits(net, bits, bits + n, bits + 2 * n, n, nullptr, logp, v, stream);
    OTH_HIP(hipFreeAsync(bits, as_stream(stream)));
    return r;
}

}  // extern "C"
