// replay_ops.hip -- section 5 of include/othello_mi355x.h: operations on replay tuples that stay on the
// device between self-play and the trainer.
#include "common.h"

namespace oth {

// get_symmetries (bitboard.pyx:338-370) for a batch of samples.  Variant k = 2j is rot90^j (numpy.rot90,
// counter-clockwise) of every plane and of pi[:64].reshape(8,8); k = 2j+1 is that followed by a left-right
// flip; pi[64] (pass) and z are copied.  Literal transform: plane 2 is the ROTATED legal mask, not the legal
// moves of the rotated position (the reference's rules are not rotation-invariant, SURVEY 8(f1) caveat).
// One wave per output sample: 3 planes + pi as 256-B coalesced rows.
template <int BS>
__device__ __forceinline__ int sym_src(int k, int r, int c) {
    constexpr int M = BS - 1;
    const int j = k >> 1;
    if (k & 1) c = M - c;
    int sr, sc;
    switch (j) {
    case 0: sr = r; sc = c; break;
    case 1: sr = c; sc = M - r; break;
    case 2: sr = M - r; sc = M - c; break;
    default: sr = M - c; sc = r; break;
    }
    return sr * BS + sc;
}

template <int BS>
__global__ __launch_bounds__(256) void k_symmetries(const float* __restrict__ st, const float* __restrict__ pi,
                                                    const float* __restrict__ z, int64_t n, float* __restrict__ st_o,
                                                    float* __restrict__ pi_o, float* __restrict__ z_o) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t o = w; o < n * 8; o += nw) {
        constexpr int CELLS = BS * BS, NPOL = CELLS + 1;
        const int64_t i = o >> 3;
        const int k = (int)(o & 7);
        if (lane < CELLS) {   // BS <= 8: one lane per square
            const int src = sym_src<BS>(k, lane / BS, lane % BS);
            const float* s = st + i * (3 * CELLS);
            float* d = st_o + o * (3 * CELLS);
            d[lane] = s[src];
            d[CELLS + lane] = s[CELLS + src];
            d[2 * CELLS + lane] = s[2 * CELLS + src];
            pi_o[o * NPOL + lane] = pi[i * NPOL + src];
        }
        if (lane == 0) {
            pi_o[o * NPOL + CELLS] = pi[i * NPOL + CELLS];
            z_o[o] = z[i];
        }
    }
}

// ReplayBuffer.sample (buffer.py:59-100) on the device: the minibatch is a gather of ring rows by index.
// One wave per sampled row: 768 B + 260 B + 4 B read and written as coalesced 256-byte segments; HBM-bound
// (2 x 1 032 B per row).  ring_size > 0: idx[i] is the LOGICAL position (0 = oldest) and the ring row is
// (start + idx[i]) % ring_size, so the caller samples without knowing where the ring currently starts.
__global__ __launch_bounds__(256) void k_replay_gather(const float* __restrict__ st, const float* __restrict__ pi,
                                                       const float* __restrict__ z, const int64_t* __restrict__ idx,
                                                       int64_t n, int64_t start, int64_t ring_size, int cells,
                                                       float* __restrict__ st_o, float* __restrict__ pi_o,
                                                       float* __restrict__ z_o) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t o = w; o < n; o += nw) {
        int64_t i = idx[o];
        if (ring_size > 0) i = (start + i) % ring_size;
        const int srow = 3 * cells, prow = cells + 1;   // 192 / 65 on 8x8, 108 / 37 on 6x6
        const float* s = st + i * srow;
        float* d = st_o + o * srow;
        for (int c = lane; c < srow; c += 64) d[c] = s[c];
        for (int c = lane; c < prow; c += 64) pi_o[o * prow + c] = pi[i * prow + c];
        if (lane == 0) z_o[o] = z[i];
    }
}

}  // namespace oth

using namespace oth;

extern "C" int oth_replay_gather_n(int board_size, const float* states, const float* pis, const float* zs,
                                   const int64_t* idx, int64_t n, int64_t ring_start, int64_t ring_size,
                                   float* states_out, float* pis_out, float* values_out, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(board_size == 8 || board_size == 6, "oth_replay_gather_n: board_size must be 8 or 6");
    OTH_CHECK(n >= 0 && ring_size >= 0 && ring_start >= 0 &&
                  (n == 0 || (states && pis && zs && idx && states_out && pis_out && values_out)),
              "oth_replay_gather: null pointer or negative size");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(states);
    int64_t blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), states, pis, zs, idx, n,
                       ring_start, ring_size, board_size * board_size, states_out, pis_out, values_out);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

extern "C" int oth_replay_gather(const float* states, const float* pis, const float* zs, const int64_t* idx, int64_t n,
                                 int64_t ring_start, int64_t ring_size, float* states_out, float* pis_out,
                                 float* values_out, void* stream) {
    return oth_replay_gather_n(8, states, pis, zs, idx, n, ring_start, ring_size, states_out, pis_out, values_out, stream);
}

extern "C" int oth_augment_symmetries_n(int board_size, const float* states, const float* pis, const float* zs,
                                        int64_t n, float* states_out, float* pis_out, float* zs_out, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(board_size == 8 || board_size == 6, "oth_augment_symmetries_n: board_size must be 8 or 6");
    OTH_CHECK(n >= 0 && (n == 0 || (states && pis && zs && states_out && pis_out && zs_out)),
              "oth_augment_symmetries: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(states);
    int64_t blocks = (n * 8 + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (board_size == 8)
        hipLaunchKernelGGL(k_symmetries<8>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), states, pis, zs, n,
                           states_out, pis_out, zs_out);
    else
        hipLaunchKernelGGL(k_symmetries<6>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), states, pis, zs, n,
                           states_out, pis_out, zs_out);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

extern "C" int oth_augment_symmetries(const float* states, const float* pis, const float* zs, int64_t n,
                                      float* states_out, float* pis_out, float* zs_out, void* stream) {
    return oth_augment_symmetries_n(8, states, pis, zs, n, states_out, pis_out, zs_out, stream);
}
