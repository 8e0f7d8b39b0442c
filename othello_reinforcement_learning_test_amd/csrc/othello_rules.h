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

constexpr uint64_t kNotA = 0xFEFEFEFEFEFEFEFEULL;  // pyx:29
constexpr uint64_t kNotH = 0x7F7F7F7F7F7F7F7FULL;  // pyx:31
constexpr uint64_t kAll = 0xFFFFFFFFFFFFFFFFULL;
constexpr uint64_t kStartSelf = (1ULL << 28) | (1ULL << 35);  // black E4,D5 (pyx:65)
constexpr uint64_t kStartOpp = (1ULL << 27) | (1ULL << 36);   // white D4,E5 (pyx:64)

// one step of a ray in direction D (pyx:20) with the reference's post-shift mask (pyx:33-38)
template <int D>
OTH_HD uint64_t step(uint64_t x) {
    if constexpr (D == 8) return x << 8;
    else if constexpr (D == -8) return x >> 8;
    else if constexpr (D == 1) return (x << 1) & kNotH;
    else if constexpr (D == -1) return (x >> 1) & kNotA;
    else if constexpr (D == 9) return (x << 9) & kNotH;
    else if constexpr (D == -9) return (x >> 9) & kNotA;
    else if constexpr (D == 7) return (x << 7) & kNotA;
    else return (x >> 7) & kNotH;  // D == -7
}
// undo n plain (unmasked) steps: the origin of a ray cell in linear index arithmetic
template <int D>
OTH_HD uint64_t back(uint64_t x, int n) {
    if constexpr (D > 0) return x >> (D * n);
    else return x << ((-D) * n);
}

template <int D>
OTH_HD uint64_t legal_dir(uint64_t self_b, uint64_t opp_b, uint64_t empty) {
    uint64_t legal = 0;
    uint64_t x = step<D>(empty) & opp_b;  // rays whose first cell is an opponent stone
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        uint64_t nxt = step<D>(x);
        legal |= back<D>(nxt & self_b, k + 1);  // bracketed: origin is k+1 steps back
        x = nxt & opp_b;
    }
    return legal;
}

// bitboard.pyx:135-158 _compute_legal_moves / :187 get_legal_moves_bits
OTH_HD uint64_t legal_moves(uint64_t self_b, uint64_t opp_b) {
    const uint64_t empty = ~(self_b | opp_b);
    uint64_t l = legal_dir<-8>(self_b, opp_b, empty) | legal_dir<8>(self_b, opp_b, empty) |
                 legal_dir<-1>(self_b, opp_b, empty) | legal_dir<1>(self_b, opp_b, empty) |
                 legal_dir<-9>(self_b, opp_b, empty) | legal_dir<-7>(self_b, opp_b, empty) |
                 legal_dir<7>(self_b, opp_b, empty) | legal_dir<9>(self_b, opp_b, empty);
    return l & empty;
}

template <int D>
OTH_HD uint64_t flip_dir(uint64_t pos_bit, uint64_t self_b, uint64_t opp_b) {  // pyx:71-114
    uint64_t flip = 0;
    uint64_t cur = step<D>(pos_bit);
#pragma unroll
    for (int k = 0; k < 7; ++k) {  // at most 7 cells fit on any ray
        const uint64_t on_opp = cur & opp_b;
        flip |= on_opp;
        cur = on_opp ? step<D>(cur) : cur;
    }
    return (cur & self_b) ? flip : 0;
}

// bitboard.pyx:116-133 _get_flip_bits (pos in 0..63)
OTH_HD uint64_t flip_bits(int pos, uint64_t self_b, uint64_t opp_b) {
    const uint64_t p = 1ULL << pos;
    return flip_dir<-8>(p, self_b, opp_b) | flip_dir<8>(p, self_b, opp_b) |
           flip_dir<-1>(p, self_b, opp_b) | flip_dir<1>(p, self_b, opp_b) |
           flip_dir<-9>(p, self_b, opp_b) | flip_dir<-7>(p, self_b, opp_b) |
           flip_dir<7>(p, self_b, opp_b) | flip_dir<9>(p, self_b, opp_b);
}

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

OTH_HD void reset(Board& b) {  // pyx:52-69
    b.self_b = kStartSelf;
    b.opp_b = kStartOpp;
    b.move_count = 0;
    b.passed = 0;
}

// bitboard.pyx:195-247 make_move.  Returns 1 on success; on failure the state is untouched.
OTH_HD int make_move(Board& b, int pos) {
    if (pos == 64) {  // pass is valid only when there is no legal move (pyx:209-219)
        if (legal_moves(b.self_b, b.opp_b) != 0) return 0;
        const uint64_t t = b.self_b;
        b.self_b = b.opp_b;
        b.opp_b = t;
        b.move_count += 1;
        b.passed = 1;
        return 1;
    }
    if (pos < 0 || pos > 63) return 0;
    const uint64_t bit = 1ULL << pos;
    if ((b.self_b | b.opp_b) & bit) return 0;
    const uint64_t flip = flip_bits(pos, b.self_b, b.opp_b);
    if (flip == 0) return 0;
    const uint64_t ns = b.opp_b & ~flip;  // swap sides (pyx:160-164)
    b.opp_b = b.self_b | bit | flip;
    b.self_b = ns;
    b.move_count += 1;
    b.passed = 0;
    return 1;
}

// move already known to be legal (search inner loop): no checks, pass when pos == 64
OTH_HD void apply_known(uint64_t& self_b, uint64_t& opp_b, int pos) {
    uint64_t flip = 0, bit = 0;
    if (pos < 64) {
        bit = 1ULL << pos;
        flip = flip_bits(pos, self_b, opp_b);
    }
    const uint64_t ns = opp_b & ~flip;
    opp_b = self_b | bit | flip;
    self_b = ns;
}

OTH_HD int is_terminal(uint64_t self_b, uint64_t opp_b) {  // pyx:249-264
    if (legal_moves(self_b, opp_b) != 0) return 0;
    return legal_moves(opp_b, self_b) == 0;
}

OTH_HD int winner(uint64_t self_b, uint64_t opp_b) {  // pyx:266-282, side-to-move relative
    const int s = popcount64(self_b), o = popcount64(opp_b);
    return s > o ? 1 : (s < o ? -1 : 0);
}

}  // namespace oth
