// rules_api.hip -- sections 1 and 2 of include/othello_mi355x.h: the host-side single-board
// functions and the batched rules kernels (one position per lane; bitboards live in VGPRs, the
// arrays are read and written fully coalesced, 8 B per lane).
#include <string.h>

#include "common.h"
#include "othello_rules.h"

namespace oth {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool device_ok() {
    static int state = -1;  // -1 unknown, 0 no, 1 yes
    if (state < 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            state = 0;
        } else {
            hipDeviceProp_t p;
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
                state = 0;
            } else {
                state = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
                if (!state) set_error("device is %s, this library is built for gfx950 only", p.gcnArchName);
            }
        }
    }
    return state == 1;
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
}

int bind_device(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == dev) return OTH_OK;
    OTH_HIP(hipSetDevice(dev));
    return OTH_OK;
}

int bind_pointer_device(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pointer %p is not a HIP allocation (device memory expected)", p);
        return OTH_E_INVALID;
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) {
        set_error("pointer %p is host memory; this call takes device pointers", p);
        return OTH_E_INVALID;
    }
    return bind_device(at.device);
}

// -------------------------------------------------------------------------------- kernels
template <int N>
__global__ void k_legal(const uint64_t* __restrict__ s, const uint64_t* __restrict__ o,
                        uint64_t* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = legal_moves_n<N>(s[i], o[i]);
}

template <int N>
__global__ void k_make_move(uint64_t* __restrict__ s, uint64_t* __restrict__ o, const int32_t* __restrict__ pos,
                            int32_t* __restrict__ ok, uint64_t* __restrict__ flips, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        Board b{s[i], o[i], 0, 0};
        const int p = pos[i];
        uint64_t f = 0;
        if (p >= 0 && p < Geo<N>::CELLS && !((b.self_b | b.opp_b) >> p & 1ULL)) f = flip_bits_n<N>(p, b.self_b, b.opp_b);
        const int r = make_move_n<N>(b, p);
        s[i] = b.self_b;
        o[i] = b.opp_b;
        ok[i] = r;
        if (flips) flips[i] = r ? f : 0;
    }
}

template <int N>
__global__ void k_status(const uint64_t* __restrict__ s, const uint64_t* __restrict__ o, int32_t* __restrict__ term,
                         int32_t* __restrict__ win, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        term[i] = is_terminal_n<N>(s[i], o[i]);
        win[i] = winner(s[i], o[i]);
    }
}

// one wave per position: lane = square, three coalesced row stores per position (256 B on 8x8)
template <int N>
__global__ void k_tensor(const uint64_t* __restrict__ s, const uint64_t* __restrict__ o, float* __restrict__ out,
                         int64_t n) {
    constexpr int CELLS = Geo<N>::CELLS;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const uint64_t sb = s[i], ob = o[i];
        const uint64_t lg = legal_moves_n<N>(sb, ob);
        float* t = out + i * 3 * CELLS;
        if (lane < CELLS) {
            t[lane] = (sb >> lane) & 1ULL ? 1.0f : 0.0f;
            t[CELLS + lane] = (ob >> lane) & 1ULL ? 1.0f : 0.0f;
            t[2 * CELLS + lane] = (lg >> lane) & 1ULL ? 1.0f : 0.0f;
        }
    }
}

// Size-independent parity property: fold legal masks / flips of an LCG position stream into two
// 64-bit accumulators.  The fold acc = acc*M + x over positions i is associative in the form
// (A, B) pairs of the affine maps, so each thread folds a contiguous chunk and a second pass folds
// the per-thread maps in order.
struct Affine {
    uint64_t mul, add;
};
__host__ __device__ inline Affine compose(Affine first, Affine then) {  // then(first(x))
    return Affine{first.mul * then.mul, first.add * then.mul + then.add};
}
__host__ __device__ inline uint64_t lcg_skip(uint64_t x, uint64_t k) {
    // advance x_{n+1} = a x_n + c by k steps
    uint64_t a = 6364136223846793005ULL, c = 1442695040888963407ULL, am = 1, cm = 0;
    while (k) {
        if (k & 1) {
            am *= a;
            cm = cm * a + c;
        }
        c = (a + 1) * c;
        a *= a;
        k >>= 1;
    }
    return am * x + cm;
}

template <int N>
__global__ void k_checksum(int64_t n, int64_t per_thread, Affine* legal_maps, Affine* flip_maps) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t lo = t * per_thread, hi = lo + per_thread < n ? lo + per_thread : n;
    Affine la{1, 0}, fa{1, 0};
    if (lo < n) {
        uint64_t x = lcg_skip(0x9E3779B97F4A7C15ULL, 3 * (uint64_t)lo);
        for (int64_t i = lo; i < hi; ++i) {
            uint64_t a, c, d;
            x = x * 6364136223846793005ULL + 1442695040888963407ULL; a = x;
            x = x * 6364136223846793005ULL + 1442695040888963407ULL; c = x;
            x = x * 6364136223846793005ULL + 1442695040888963407ULL; d = x;
            const uint64_t occ = (i & 1) ? (a | (c & d)) : (a & c);
            const uint64_t sb = occ & d & Geo<N>::all(), ob = occ & ~d & Geo<N>::all();
            const uint64_t lb = legal_moves_n<N>(sb, ob);
            la = compose(la, Affine{0x100000001B3ULL, lb});
            if (lb) {
                const int mv = __ffsll((unsigned long long)lb) - 1;
                fa = compose(fa, Affine{0x100000001B3ULL, flip_bits_n<N>(mv, sb, ob)});
            }
        }
    }
    legal_maps[t] = la;
    flip_maps[t] = fa;
}

static inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g > 2048) g = 2048;  // cap and grid-stride (memory-bound kernels)
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace oth

using namespace oth;

extern "C" {

const char* oth_last_error(void) { return g_err; }
int oth_device_available(void) { return device_ok() ? 1 : 0; }
const char* oth_version(void) { return "othello_mi355x 0.1 (gfx950)"; }

// ---- section 1: host single-board API ----------------------------------------------------------
void oth_board_reset(oth_board* b) {
    Board t;
    reset(t);
    b->self_board = t.self_b; b->opp_board = t.opp_b; b->move_count = 0; b->passed = 0;
}
uint64_t oth_legal_moves(uint64_t s, uint64_t o) { return legal_moves(s, o); }
uint64_t oth_flip_bits(int pos, uint64_t s, uint64_t o) { return (pos < 0 || pos > 63) ? 0 : flip_bits(pos, s, o); }
int oth_board_make_move(oth_board* b, int pos) {
    Board t{b->self_board, b->opp_board, b->move_count, b->passed};
    const int r = make_move(t, pos);
    b->self_board = t.self_b; b->opp_board = t.opp_b; b->move_count = t.move_count; b->passed = t.passed;
    return r;
}
int oth_board_is_terminal(const oth_board* b) { return is_terminal(b->self_board, b->opp_board); }
int oth_board_get_winner(const oth_board* b) { return winner(b->self_board, b->opp_board); }
void oth_board_get_tensor_input(const oth_board* b, float* t) {
    const uint64_t lg = legal_moves(b->self_board, b->opp_board);
    for (int i = 0; i < 64; ++i) {
        t[i] = (b->self_board >> i) & 1ULL ? 1.0f : 0.0f;
        t[64 + i] = (b->opp_board >> i) & 1ULL ? 1.0f : 0.0f;
        t[128 + i] = (lg >> i) & 1ULL ? 1.0f : 0.0f;
    }
}
// get_symmetries (bitboard.pyx:338-370): element k=2j is rot90^j (counter-clockwise, numpy.rot90),
// k=2j+1 is that followed by a left-right flip; pi[64] (pass) is carried over unchanged.
static inline int sym_src(int k, int r, int c) {
    const int j = k >> 1;
    if (k & 1) c = 7 - c;  // undo the flip first: out[r][c] = rot[r][7-c]
    int sr, sc;
    switch (j) {
    case 0: sr = r; sc = c; break;
    case 1: sr = c; sc = 7 - r; break;
    case 2: sr = 7 - r; sc = 7 - c; break;
    default: sr = 7 - c; sc = r; break;
    }
    return sr * 8 + sc;
}
void oth_board_get_symmetries(const oth_board* b, const float* pi, float* states, float* pis) {
    float t[192];
    oth_board_get_tensor_input(b, t);
    for (int k = 0; k < 8; ++k)
        for (int r = 0; r < 8; ++r)
            for (int c = 0; c < 8; ++c) {
                const int src = sym_src(k, r, c), dst = r * 8 + c;
                for (int ch = 0; ch < 3; ++ch) states[k * 192 + ch * 64 + dst] = t[ch * 64 + src];
                pis[k * 65 + dst] = pi[src];
            }
    for (int k = 0; k < 8; ++k) pis[k * 65 + 64] = pi[64];
}

// ---- section 2: batched device rules -----------------------------------------------------------
#define OTH_BOARD_OK(bs, what) OTH_CHECK((bs) == 8 || (bs) == 6, what ": board_size must be 8 or 6")
int oth_legal_moves_batch_n(int bs, const uint64_t* s, const uint64_t* o, uint64_t* legal, int64_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_BOARD_OK(bs, "oth_legal_moves_batch");
    OTH_CHECK(n >= 0 && (n == 0 || (s && o && legal)), "oth_legal_moves_batch: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(s);
    if (bs == 8) hipLaunchKernelGGL(k_legal<8>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, legal, n);
    else hipLaunchKernelGGL(k_legal<6>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, legal, n);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}
int oth_make_move_batch_n(int bs, uint64_t* s, uint64_t* o, const int32_t* pos, int32_t* ok, uint64_t* flips, int64_t n,
                          void* stream) {
    OTH_NEED_DEVICE();
    OTH_BOARD_OK(bs, "oth_make_move_batch");
    OTH_CHECK(n >= 0 && (n == 0 || (s && o && pos && ok)), "oth_make_move_batch: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(s);
    if (bs == 8) hipLaunchKernelGGL(k_make_move<8>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, pos, ok, flips, n);
    else hipLaunchKernelGGL(k_make_move<6>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, pos, ok, flips, n);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}
int oth_status_batch_n(int bs, const uint64_t* s, const uint64_t* o, int32_t* term, int32_t* win, int64_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_BOARD_OK(bs, "oth_status_batch");
    OTH_CHECK(n >= 0 && (n == 0 || (s && o && term && win)), "oth_status_batch: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(s);
    if (bs == 8) hipLaunchKernelGGL(k_status<8>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, term, win, n);
    else hipLaunchKernelGGL(k_status<6>, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), s, o, term, win, n);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}
int oth_tensor_input_batch_n(int bs, const uint64_t* s, const uint64_t* o, float* out, int64_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_BOARD_OK(bs, "oth_tensor_input_batch");
    OTH_CHECK(n >= 0 && (n == 0 || (s && o && out)), "oth_tensor_input_batch: null pointer or negative n");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(s);
    const dim3 grid(grid_for((n + 3) / 4 * 256, 256));
    if (bs == 8) hipLaunchKernelGGL(k_tensor<8>, grid, dim3(256), 0, as_stream(stream), s, o, out, n);
    else hipLaunchKernelGGL(k_tensor<6>, grid, dim3(256), 0, as_stream(stream), s, o, out, n);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}
int oth_rules_checksum_n(int bs, int64_t n, uint64_t* legal_acc, uint64_t* flip_acc, void* stream) {
    OTH_NEED_DEVICE();
    OTH_BOARD_OK(bs, "oth_rules_checksum");
    OTH_CHECK(n >= 0 && legal_acc && flip_acc, "oth_rules_checksum: bad arguments");
    const int threads = 256 * 1024;
    const int64_t per = (n + threads - 1) / threads > 0 ? (n + threads - 1) / threads : 1;
    Affine *dl = nullptr, *df = nullptr;
    OTH_HIP(hipMalloc(&dl, sizeof(Affine) * threads));
    OTH_HIP(hipMalloc(&df, sizeof(Affine) * threads));
    if (bs == 8) hipLaunchKernelGGL(k_checksum<8>, dim3(threads / 256), dim3(256), 0, as_stream(stream), n, per, dl, df);
    else hipLaunchKernelGGL(k_checksum<6>, dim3(threads / 256), dim3(256), 0, as_stream(stream), n, per, dl, df);
    Affine* hl = new Affine[threads];
    Affine* hf = new Affine[threads];
    hipError_t e1 = hipMemcpyAsync(hl, dl, sizeof(Affine) * threads, hipMemcpyDeviceToHost, as_stream(stream));
    hipError_t e2 = hipMemcpyAsync(hf, df, sizeof(Affine) * threads, hipMemcpyDeviceToHost, as_stream(stream));
    hipError_t e3 = hipStreamSynchronize(as_stream(stream));
    uint64_t la = 0, fa = 0;
    for (int t = 0; t < threads; ++t) {
        la = la * hl[t].mul + hl[t].add;
        fa = fa * hf[t].mul + hf[t].add;
    }
    delete[] hl;
    delete[] hf;
    (void)hipFree(dl);
    (void)hipFree(df);
    OTH_HIP(e1);
    OTH_HIP(e2);
    OTH_HIP(e3);
    *legal_acc = la;
    *flip_acc = fa;
    return OTH_OK;
}
// the reference's board (8x8): the original entry points
int oth_legal_moves_batch(const uint64_t* s, const uint64_t* o, uint64_t* legal, int64_t n, void* stream) {
    return oth_legal_moves_batch_n(8, s, o, legal, n, stream);
}
int oth_make_move_batch(uint64_t* s, uint64_t* o, const int32_t* pos, int32_t* ok, uint64_t* flips, int64_t n,
                        void* stream) {
    return oth_make_move_batch_n(8, s, o, pos, ok, flips, n, stream);
}
int oth_status_batch(const uint64_t* s, const uint64_t* o, int32_t* term, int32_t* win, int64_t n, void* stream) {
    return oth_status_batch_n(8, s, o, term, win, n, stream);
}
int oth_tensor_input_batch(const uint64_t* s, const uint64_t* o, float* out, int64_t n, void* stream) {
    return oth_tensor_input_batch_n(8, s, o, out, n, stream);
}
int oth_rules_checksum(int64_t n, uint64_t* legal_acc, uint64_t* flip_acc, void* stream) {
    return oth_rules_checksum_n(8, n, legal_acc, flip_acc, stream);
}

// ---- 6x6 host functions (single board): same header, N = 6 ------------------------------------------------------
void oth_board_reset_n(int bs, oth_board* b) {
    Board t;
    if (bs == 6) reset_n<6>(t); else reset_n<8>(t);
    b->self_board = t.self_b; b->opp_board = t.opp_b; b->move_count = 0; b->passed = 0;
}
uint64_t oth_legal_moves_n(int bs, uint64_t s, uint64_t o) { return bs == 6 ? legal_moves_n<6>(s, o) : legal_moves_n<8>(s, o); }
uint64_t oth_flip_bits_n(int bs, int pos, uint64_t s, uint64_t o) {
    if (pos < 0 || pos >= bs * bs) return 0;
    return bs == 6 ? flip_bits_n<6>(pos, s, o) : flip_bits_n<8>(pos, s, o);
}
int oth_board_make_move_n(int bs, oth_board* b, int pos) {
    Board t{b->self_board, b->opp_board, b->move_count, b->passed};
    const int r = bs == 6 ? make_move_n<6>(t, pos) : make_move_n<8>(t, pos);
    b->self_board = t.self_b; b->opp_board = t.opp_b; b->move_count = t.move_count; b->passed = t.passed;
    return r;
}
int oth_board_is_terminal_n(int bs, const oth_board* b) {
    return bs == 6 ? is_terminal_n<6>(b->self_board, b->opp_board) : is_terminal_n<8>(b->self_board, b->opp_board);
}

}  // extern "C"
