// othello_rules.h -- bitboard rules as inline host+device functions (gfx950 device code and the
// host-side single-board API of the C-ABI share this one source).
//
// Semantics are those of /root/reference/src/cython/bitboard.pyx INCLUDING its edge behaviour
// (SURVEY.md L2): the A/H-file masks are applied AFTER the shift, so the col-1 rays (-1,-9,+7)
// die on landing in file A and wrap A->H, the col+1 rays (+1,-7,+9) die on landing in file H and
// wrap H->A; the +-8 rays are plain.  Bit-exact parity with that is the spec, not real Othello.
//
// Unlike the reference (one ray walk per empty square, pyx:148-158) legal-move generation here is
// set-wise: all 64 candidate origins advance together, 6 steps per direction (a ray can hold at
// most 6 opponent stones before it must land on an own stone; verified against the oracle).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define OTH_HD __host__ __device__ __forceinline__
#else
#define OTH_HD static inline
#endif

namespace oth {

// Board geometry.  N = 8 is the reference's game: every constant below then equals bitboard.pyx:20-38,62-69.
// N = 6 is the same algorithm on a 6x6 grid (BASELINE configs[4]; reference configs/debug_6x6.yaml names the size
// but the reference has NO 6x6 rules -- its game.size is never read): bit i = row*6 + col, pass action 36, the same
// eight directions with the reference's post-shift masks carried over (column 0 cleared where the reference clears
// file A, column N-1 where it clears file H, i.e. the same swapped-mask behaviour: the col-1 rays die on landing in
// column 0 and wrap 0 -> N-1, the col+1 rays die on landing in column N-1 and wrap N-1 -> 0); bits >= 36 never
// survive a step.  PARITY UNPINNED for N = 6 (checked against oracle/libothello_oracle6.so, the same definition).
template <int N>
struct Geo {
    static constexpr int CELLS = N * N;
    static constexpr int NPOL = CELLS + 1;  // policy length: squares + pass
    static constexpr uint64_t all() { return N == 8 ? ~0ULL : ((1ULL << (N * N)) - 1ULL); }
    static constexpr uint64_t col(int c) {
        uint64_t m = 0;
        for (int r = 0; r < N; ++r) m |= 1ULL << (r * N + c);
        return m;
    }
    static constexpr uint64_t not_a() { return all() & ~col(0); }      // pyx:29 NOT_A_FILE
    static constexpr uint64_t not_h() { return all() & ~col(N - 1); }  // pyx:31 NOT_H_FILE
    static constexpr uint64_t start_self() {  // black: (N/2-1, N/2), (N/2, N/2-1)   (pyx:65: bits 28, 35)
        return (1ULL << ((N / 2 - 1) * N + N / 2)) | (1ULL << ((N / 2) * N + N / 2 - 1));
    }
    static constexpr uint64_t start_opp() {   // white: (N/2-1, N/2-1), (N/2, N/2)   (pyx:64: bits 27, 36)
        return (1ULL << ((N / 2 - 1) * N + N / 2 - 1)) | (1ULL << ((N / 2) * N + N / 2));
    }
    // direction K of pyx:20 DIRECTIONS = [-N, +N, -1, +1, -(N+1), -(N-1), +(N-1), +(N+1)]
    static constexpr int delta(int k) {
        return k == 0 ? -N : k == 1 ? N : k == 2 ? -1 : k == 3 ? 1 : k == 4 ? -(N + 1) : k == 5 ? -(N - 1)
               : k == 6 ? (N - 1) : (N + 1);
    }
    // ... and its mask of pyx:33-38: [ALL, ALL, NOT_A, NOT_H, NOT_A, NOT_H, NOT_A, NOT_H]
    static constexpr uint64_t mask(int k) { return k < 2 ? all() : ((k & 1) ? not_h() : not_a()); }
};
static_assert(Geo<8>::not_a() == 0xFEFEFEFEFEFEFEFEULL && Geo<8>::not_h() == 0x7F7F7F7F7F7F7F7FULL, "pyx:29,31");
static_assert(Geo<8>::start_self() == ((1ULL << 28) | (1ULL << 35)) && Geo<8>::start_opp() == ((1ULL << 27) | (1ULL << 36)),
              "pyx:64-65");
static_assert(Geo<6>::all() == 0xFFFFFFFFFULL && Geo<6>::not_a() == 0xFBEFBEFBEULL && Geo<6>::not_h() == 0x7DF7DF7DFULL, "6x6 masks");

constexpr uint64_t kNotA = Geo<8>::not_a();
constexpr uint64_t kNotH = Geo<8>::not_h();
constexpr uint64_t kAll = Geo<8>::all();
constexpr uint64_t kStartSelf = Geo<8>::start_self();  // black E4,D5 (pyx:65)
constexpr uint64_t kStartOpp = Geo<8>::start_opp();    // white D4,E5 (pyx:64)

// one step of a ray in direction K with the reference's post-shift mask (pyx:33-38)
template <int N, int K>
OTH_HD uint64_t step(uint64_t x) {
    constexpr int d = Geo<N>::delta(K);
    constexpr uint64_t m = Geo<N>::mask(K);
    if constexpr (d > 0) return (x << d) & m;
    else return (x >> (-d)) & m;
}
// undo n plain (unmasked) steps: the origin of a ray cell in linear index arithmetic
template <int N, int K>
OTH_HD uint64_t back(uint64_t x, int n) {
    constexpr int d = Geo<N>::delta(K);
    if constexpr (d > 0) return x >> (d * n);
    else return x << ((-d) * n);
}

// Set-wise legal-move generation: all candidate origins advance together, N-2 steps per direction (a ray can hold
// at most N-2 opponent stones before it must land on an own stone; for N = 8 verified against the compiled
// reference on 200 799 positions, for N = 6 against the 6x6 oracle).
template <int N, int K>
OTH_HD uint64_t legal_dir(uint64_t self_b, uint64_t opp_b, uint64_t empty) {
    uint64_t legal = 0;
    uint64_t x = step<N, K>(empty) & opp_b;  // rays whose first cell is an opponent stone
#pragma unroll
    for (int k = 1; k <= N - 2; ++k) {
        uint64_t nxt = step<N, K>(x);
        legal |= back<N, K>(nxt & self_b, k + 1);  // bracketed: origin is k+1 steps back
        x = nxt & opp_b;
    }
    return legal;
}

// bitboard.pyx:135-158 _compute_legal_moves / :187 get_legal_moves_bits
template <int N>
OTH_HD uint64_t legal_moves_n(uint64_t self_b, uint64_t opp_b) {
    const uint64_t empty = ~(self_b | opp_b) & Geo<N>::all();
    uint64_t l = legal_dir<N, 0>(self_b, opp_b, empty) | legal_dir<N, 1>(self_b, opp_b, empty) |
                 legal_dir<N, 2>(self_b, opp_b, empty) | legal_dir<N, 3>(self_b, opp_b, empty) |
                 legal_dir<N, 4>(self_b, opp_b, empty) | legal_dir<N, 5>(self_b, opp_b, empty) |
                 legal_dir<N, 6>(self_b, opp_b, empty) | legal_dir<N, 7>(self_b, opp_b, empty);
    return l & empty;
}
OTH_HD uint64_t legal_moves(uint64_t self_b, uint64_t opp_b) { return legal_moves_n<8>(self_b, opp_b); }

template <int N, int K>
OTH_HD uint64_t flip_dir(uint64_t pos_bit, uint64_t self_b, uint64_t opp_b) {  // pyx:71-114
    uint64_t flip = 0;
    uint64_t cur = step<N, K>(pos_bit);
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {  // at most N-1 cells fit on any ray
        const uint64_t on_opp = cur & opp_b;
        flip |= on_opp;
        cur = on_opp ? step<N, K>(cur) : cur;
    }
    return (cur & self_b) ? flip : 0;
}

// bitboard.pyx:116-133 _get_flip_bits (pos in 0..N*N-1)
template <int N>
OTH_HD uint64_t flip_bits_n(int pos, uint64_t self_b, uint64_t opp_b) {
    const uint64_t p = 1ULL << pos;
    return flip_dir<N, 0>(p, self_b, opp_b) | flip_dir<N, 1>(p, self_b, opp_b) | flip_dir<N, 2>(p, self_b, opp_b) |
           flip_dir<N, 3>(p, self_b, opp_b) | flip_dir<N, 4>(p, self_b, opp_b) | flip_dir<N, 5>(p, self_b, opp_b) |
           flip_dir<N, 6>(p, self_b, opp_b) | flip_dir<N, 7>(p, self_b, opp_b);
}
OTH_HD uint64_t flip_bits(int pos, uint64_t self_b, uint64_t opp_b) { return flip_bits_n<8>(pos, self_b, opp_b); }

OTH_HD int popcount64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

struct Board {  // bitboard.pxd:25-28
    uint64_t self_b, opp_b;
    int32_t move_count;
    int32_t passed;
};

template <int N>
OTH_HD void reset_n(Board& b) {  // pyx:52-69
    b.self_b = Geo<N>::start_self();
    b.opp_b = Geo<N>::start_opp();
    b.move_count = 0;
    b.passed = 0;
}
OTH_HD void reset(Board& b) { reset_n<8>(b); }

// bitboard.pyx:195-247 make_move.  Returns 1 on success; on failure the state is untouched.
template <int N>
OTH_HD int make_move_n(Board& b, int pos) {
    if (pos == Geo<N>::CELLS) {  // pass is valid only when there is no legal move (pyx:209-219)
        if (legal_moves_n<N>(b.self_b, b.opp_b) != 0) return 0;
        const uint64_t t = b.self_b;
        b.self_b = b.opp_b;
        b.opp_b = t;
        b.move_count += 1;
        b.passed = 1;
        return 1;
    }
    if (pos < 0 || pos > Geo<N>::CELLS - 1) return 0;
    const uint64_t bit = 1ULL << pos;
    if ((b.self_b | b.opp_b) & bit) return 0;
    const uint64_t flip = flip_bits_n<N>(pos, b.self_b, b.opp_b);
    if (flip == 0) return 0;
    const uint64_t ns = b.opp_b & ~flip;  // swap sides (pyx:160-164)
    b.opp_b = b.self_b | bit | flip;
    b.self_b = ns;
    b.move_count += 1;
    b.passed = 0;
    return 1;
}
OTH_HD int make_move(Board& b, int pos) { return make_move_n<8>(b, pos); }

// move already known to be legal (search inner loop): no checks, pass when pos == N*N
template <int N>
OTH_HD void apply_known_n(uint64_t& self_b, uint64_t& opp_b, int pos) {
    uint64_t flip = 0, bit = 0;
    if (pos < Geo<N>::CELLS) {
        bit = 1ULL << pos;
        flip = flip_bits_n<N>(pos, self_b, opp_b);
    }
    const uint64_t ns = opp_b & ~flip;
    opp_b = self_b | bit | flip;
    self_b = ns;
}
OTH_HD void apply_known(uint64_t& self_b, uint64_t& opp_b, int pos) { apply_known_n<8>(self_b, opp_b, pos); }

template <int N>
OTH_HD int is_terminal_n(uint64_t self_b, uint64_t opp_b) {  // pyx:249-264
    if (legal_moves_n<N>(self_b, opp_b) != 0) return 0;
    return legal_moves_n<N>(opp_b, self_b) == 0;
}
OTH_HD int is_terminal(uint64_t self_b, uint64_t opp_b) { return is_terminal_n<8>(self_b, opp_b); }

OTH_HD int winner(uint64_t self_b, uint64_t opp_b) {  // pyx:266-282, side-to-move relative
    const int s = popcount64(self_b), o = popcount64(opp_b);
    return s > o ? 1 : (s < o ? -1 : 0);
}

}  // namespace oth
