// engine.hip -- section 4 of include/othello_mi355x.h: PUCT tree search and self-play on the device.
//
// Execution model (gfx950): ONE WAVEFRONT PER GAME, LANE = BOARD SQUARE.  A node's children are
// its legal squares in ascending order (reference node.py:83-87 inserts them in that order), so the
// 64 lanes of a wave score all children of a node in one pass (float64, as the reference's numpy
// arithmetic does) and a ballot picks the first maximum.  Bitboards stay in registers while a wave
// descends; the tree (32-B nodes, 16-B edges) lives in a per-game arena in HBM that is small
// enough to stay L2/Infinity-Cache resident.  Leaves are appended to a dense evaluation batch
// (24 B per position: self, opp, legal) that the network kernel consumes without a host round trip.
//
// ONE tree kernel per simulation (k_tree): expand the leaf evaluated by the previous network launch
// and back its value up (node.py:62-89, mcts.py:152-168), then -- same wave, same launch -- select the
// next leaf (node.py:91-126) or, after the last simulation, finish the ply (pi, move, end of game,
// slot refill, next root).  The evaluation-batch cursor is double-buffered (the launch that fills one
// counter zeroes the other), so a simulation costs exactly two launches: k_tree and the network.
//
// Reference semantics reproduced: L8 expand, L9 select, L10 backup incl. the root
// never being backed up, L11 terminal leaves, L13 policy from visit counts, L14-L18 worker loops.
#include <limits.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "net.h"
#include "othello_rules.h"
#include "wave_bfly.h"

namespace oth {

constexpr int kMaxPly = 128;  // hard bound: <=60 stones placed, no two consecutive passes => <=121 plies
constexpr int kEdgesPerNode = 60;  // children <= number of empty squares <= 60

struct __attribute__((aligned(16))) Node {  // 32 B
    uint64_t self_b, opp_b, legal;
    uint32_t edge_base;
    uint32_t n_children;
};
struct __attribute__((aligned(16))) Edge {  // 16 B: one 128-bit load per lane
    double w;        // value_sum, python float (node.py:39)
    float prior;     // np.float32 (node.py:84)
    uint16_t n;      // visit_count (node.py:38)
    uint16_t child;  // node index once expanded, 0 = not expanded (the root is never a child)
};

enum : int32_t { PEND_NONE = 0, PEND_ROOT = 1, PEND_LEAF = 2 };
enum : int { TREE_NONE = 0, TREE_SELECT = 1, TREE_PLY = 2 };       // what k_tree does after the expansion
enum : int { ST_ACTIVE = 0, ST_DONE = 1, ST_FLAGS = 2, ST_NEXT = 3 };  // words of Dev::status
enum : int32_t { FLAG_HIST_OVERFLOW = 1 };
constexpr int kCachedSlot = -2;  // eval_slot value: the result row is in Dev::cres[g] (evaluation-cache hit)

struct Dev {  // device pointers + scalars handed to every kernel by value
    int32_t n_slots, cap_nodes, cap_edges, cap_path, num_sims, temp_threshold, store_late_onehot;
    float c_puct;
    // per slot
    uint64_t *g_self, *g_opp;        // current game position
    int32_t *g_ply, *g_id, *g_active;  // ply counter, game id, slot searching this step
    int32_t* g_join;                 // round at which an idle slot starts its first game (-1: never)
    Node* nodes;
    Edge* edges;
    int32_t *n_nodes, *n_edges;
    uint32_t* path;
    int32_t *path_len, *pend, *eval_slot;
    uint64_t *leaf_self, *leaf_opp, *leaf_legal;
    // evaluation batch (dense, filled through an atomic cursor; the cursor is double-buffered)
    uint64_t *ev_self, *ev_opp, *ev_legal;
    int32_t *n_eval, *n_eval_next;   // cursor of THIS launch, cursor of the next one (zeroed by this launch)
    float *logp, *val;
    const double* sqrt_tab;  // sqrt(n) exactly as numpy computes it, n = 0..num_sims+1
    // self-play bookkeeping
    int32_t* status;         // [ST_ACTIVE] playing slots, [ST_DONE] finished games, [ST_FLAGS], [ST_NEXT] next game id
    int32_t game_limit;      // ids >= game_limit are not started
    int32_t hist_mask;       // history ring: game id & hist_mask
    int32_t round;           // ply round of this launch (0 = initial joins)
    uint64_t* hist_bits;     // [hist_cap][kMaxPly][3]
    float* hist_pi;          // [hist_cap][kMaxPly][65]
    int32_t *game_len, *game_winner;  // [hist_cap]; game_len: -1 free, 0 in progress, >0 finished (plies)
    int32_t* done_list;      // [hist_cap] ring of finished game ids in completion order
    unsigned long long* counters;  // [blocks][8]
    uint64_t seed;
    // optional evaluation cache (transposition table of network outputs): C entries = C/2 sets of TWO ways (round 5: a set's
    // ways are entries 2s and 2s+1; an insert takes an empty way, else replaces the way inserted longer ago)
    uint64_t* ck;       // [C][2] keys (self, opp); all-ones = empty
    float* cv;          // [C][66] raw policy (65) + value
    int32_t* clk;       // [C] claim epoch of the last insert (also the age of the entry: FIFO within a set)
    float* cres;        // [n_slots][66] result row copied at lookup time (an insert may replace the entry later)
    uint32_t cmask;     // C - 1, 0 = cache disabled
    int32_t cepoch;
    // cache statistics (oth_engine_cache_stats): a bitmap of 64 bits per entry over a second hash of the position -- "has
    // this position been evaluated since the cache was cleared?" -- splits the misses into first evaluations (compulsory)
    // and repeats (the entry was replaced in between, or the same position missed twice in one launch); per-block rows
    unsigned long long* cseen;       // [C] words of 64 bits
    unsigned long long* cstat;       // [blocks][4]: distinct positions, repeated evaluations, conflict evictions, -
};

// ---- wave helpers --------------------------------------------------------------------------------
// (xor butterflies without the LDS crossbar: wave_bfly.h; same pairing as the __shfl_xor loops they replaced, same results)
__device__ __forceinline__ double wave_max_f64(double v) { return bfly_max_f64(v); }
__device__ __forceinline__ int wave_sum_i32(int v) { return bfly_sum_i32(v); }
__device__ __forceinline__ int wave_max_i32(int v) { return bfly_max_i32(v); }

// mcts.py:152-168 / parallel_self_play.py:199-204: child.update(v); v = -v from the leaf upwards.
// Edges on a path are distinct, so lane i updates path entry i (sign by distance from the leaf).
// link_child != 0: the leaf edge (path[depth-1]) additionally gets its new child node id, in the same
// read-modify-write (no second store to that edge from another lane).
template <typename PathT>
__device__ __forceinline__ void backup_path(Edge* edges, const PathT* path, int depth, double value, int lane,
                                            int link_child) {
    for (int i = lane; i < depth; i += 64) {
        Edge* e = &edges[path[i]];
        const double v = ((depth - 1 - i) & 1) ? -value : value;
        Edge t = *e;
        t.n = (uint16_t)(t.n + 1);
        t.w = t.w + v;
        if (link_child && i == depth - 1) t.child = (uint16_t)link_child;
        *e = t;
    }
}

// Evaluation-batch slots are handed out per BLOCK: the 4 games of a block post whether they need a
// slot, thread 0 does ONE atomicAdd for the block (4096 same-address atomics per launch serialised
// to ~90 us; 1024 take ~12 us) and the waves take consecutive slots.  Event counters are kept per block
// (plain adds to the block's own row, summed on the host at the end of a run).
// Every thread of the block must call this (it contains barriers).
struct BlockTally {
    int need[4];
    int base;
};
__device__ __forceinline__ int block_alloc_eval(const Dev& d, BlockTally& bt, bool need, int n_sims, int n_term,
                                                int n_plies, int n_games, int n_hits = 0) {
    __shared__ int s_cnt[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        bt.need[wave] = need ? 1 : 0;
        s_cnt[wave][0] = n_sims; s_cnt[wave][1] = n_term; s_cnt[wave][2] = n_plies; s_cnt[wave][3] = n_games;
        s_cnt[wave][4] = n_hits;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = bt.need[0] + bt.need[1] + bt.need[2] + bt.need[3];
        bt.base = total ? atomicAdd(d.n_eval, total) : 0;
        unsigned long long* c = d.counters + (size_t)blockIdx.x * 8;
        c[0] += (unsigned long long)total;
        c[1] += (unsigned long long)(s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0]);
        c[5] += (unsigned long long)(s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1]);
        c[2] += (unsigned long long)(s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2]);
        c[3] += (unsigned long long)(s_cnt[0][3] + s_cnt[1][3] + s_cnt[2][3] + s_cnt[3][3]);
        c[6] += (unsigned long long)(s_cnt[0][4] + s_cnt[1][4] + s_cnt[2][4] + s_cnt[3][4]);
    }
    __syncthreads();
    int slot = bt.base;
    for (int w = 0; w < wave; ++w) slot += bt.need[w];
    return slot;
}

// Evaluation cache lookup: the network is a pure function of (self, opp), so an earlier result for the
// same position (previous ply's subtree, a transposition, another game) can be reused bit for bit.
__device__ __forceinline__ uint32_t cache_slot(const Dev& d, uint64_t sb, uint64_t ob) {
    uint64_t h = ob + 0x9E3779B97F4A7C15ULL;
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ULL;
    h = (h ^ (h >> 27)) * 0x94D049BB133111EBULL;
    h = (h ^ (h >> 31)) ^ sb;
    h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ULL;
    h = (h ^ (h >> 27)) * 0x94D049BB133111EBULL;
    return (uint32_t)(h ^ (h >> 31)) & d.cmask & ~1u;   // way 0 of the position's set (way 1 = + 1)
}
// On a hit the 66-float result row is copied into the game's own row cres[g] right away: the entry may be
// replaced by an insert before the expansion (next launch) reads it.  Whole wave calls; returns hit or not.
__device__ __forceinline__ bool cache_fetch(const Dev& d, int g, uint64_t sb, uint64_t ob, int lane) {
    if (d.cmask == 0) return false;
    uint32_t cs = cache_slot(d, sb, ob);
    const ulonglong2 k0 = *(const ulonglong2*)(d.ck + 2 * (size_t)cs), k1 = *(const ulonglong2*)(d.ck + 2 * (size_t)cs + 2);
    if (k0.x == sb && k0.y == ob) {
    } else if (k1.x == sb && k1.y == ob) {
        cs += 1;
    } else {
        return false;
    }
    const float* src = d.cv + (size_t)cs * 66;
    float* dst = d.cres + (size_t)g * 66;   // [0, NPOL) policy, [65] value (row layout shared by both board sizes)
    dst[lane] = src[lane];
    if (lane < 2) dst[64 + lane] = src[64 + lane];
    return true;
}

__device__ __forceinline__ void write_eval(const Dev& d, int slot, uint64_t sb, uint64_t ob, uint64_t lg, int lane) {
    if (lane == 0) {
        d.ev_self[slot] = sb;
        d.ev_opp[slot] = ob;
        d.ev_legal[slot] = lg;
    }
}

// ---- expand + backup (node.py:62-89, mcts.py:133-148) ---------------------------------------------
// The pending position of game g (its root, or the leaf chosen by the previous launch) has been evaluated:
// mask + renormalise the priors, append the node and its edges, link it and back the value up.
template <int BS>
__device__ __forceinline__ void expand_pending(const Dev& d, int g, int lane, int pend,
                                               const float* __restrict__ policy, const float* __restrict__ value,
                                               int is_log) {
    constexpr int CELLS = Geo<BS>::CELLS, NP = Geo<BS>::NPOL;
    const int slot = d.eval_slot[g];
    const float* pol = slot >= 0 ? policy + (size_t)slot * NP : d.cres + (size_t)g * 66;
    const uint64_t sb = d.leaf_self[g], ob = d.leaf_opp[g], legal = d.leaf_legal[g];
    const bool pass = legal == 0;
    float p = lane < CELLS ? pol[lane] : -INFINITY, p64 = pol[CELLS];
    if (is_log) {  // policy_probs = torch.exp(policy_logits), mcts.py:189
        p = expf(p);
        p64 = expf(p64);
    }
    // masked_probs[legal] = policy_probs[legal] (node.py:71-72); element CELLS (64 on 8x8) is the pass action
    const bool act = pass ? false : ((legal >> lane) & 1ULL);
    const float m = act ? p : 0.0f;
    const float m64 = pass ? p64 : 0.0f;
    // prob_sum = masked_probs.sum(): numpy float32 pairwise reduction over NP elements (65 on 8x8), i.e.
    // r[j] = a[j] + a[8+j] + ... over the NP/8 full groups of 8 (in that order),
    // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the NP % 8 trailing elements one by one (8x8: a[64] only)
    float r = m;
#pragma unroll
    for (int i = 1; i < NP / 8; ++i) r += __shfl(m, (8 * i + lane) & 63);
    const float r0 = __shfl(r, 0), r1 = __shfl(r, 1), r2 = __shfl(r, 2), r3 = __shfl(r, 3);
    const float r4 = __shfl(r, 4), r5 = __shfl(r, 5), r6 = __shfl(r, 6), r7 = __shfl(r, 7);
    float sum = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
#pragma unroll
    for (int i = 8 * (NP / 8); i < CELLS; ++i) sum += __shfl(m, i);
    sum += m64;
    const int nch = pass ? 1 : __popcll(legal);
    float prior;
    if (sum > 0.0f) prior = (pass ? m64 : m) / sum;                  // masked_probs /= prob_sum
    else prior = (float)(1.0 / (double)nch);                         // node.py:78-80 uniform fallback
    Node* nodes = d.nodes + (size_t)g * d.cap_nodes;
    Edge* edges = d.edges + (size_t)g * d.cap_edges;
    const int id = d.n_nodes[g], base = d.n_edges[g];
    const bool wr = pass ? (lane == 0) : act;
    if (wr) {
        const int rank = __popcll(legal & ((1ULL << lane) - 1ULL));
        Edge e;
        e.w = 0.0; e.prior = prior; e.n = 0; e.child = 0;
        edges[base + rank] = e;
    }
    const int depth = pend == PEND_LEAF ? d.path_len[g] : 0;
    const uint32_t* path = d.path + (size_t)g * d.cap_path;
    if (lane == 0) {
        Node nd;
        nd.self_b = sb; nd.opp_b = ob; nd.legal = legal; nd.edge_base = (uint32_t)base; nd.n_children = (uint32_t)nch;
        nodes[id] = nd;
        d.n_nodes[g] = id + 1;
        d.n_edges[g] = base + nch;
        d.pend[g] = PEND_NONE;
    }
    if (depth > 0) {
        const double v = (double)(slot >= 0 ? value[slot] : pol[65]);  // values[j].item(): float32 -> python float (cache rows: [65])
        backup_path(edges, path, depth, v, lane, id);
    }
}

// Insert the fresh network results of the pending positions into the evaluation cache.  Runs as its own
// launch between the network and k_tree, so no entry is read (cache_fetch, in k_tree) while it is rewritten.
template <int BS>
__global__ __launch_bounds__(256) void k_cache_insert(Dev d, const float* __restrict__ policy,
                                                      const float* __restrict__ value) {
    constexpr int CELLS = Geo<BS>::CELLS, NP = Geo<BS>::NPOL;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= d.n_slots || d.pend[g] == PEND_NONE) return;
    const int slot = d.eval_slot[g];
    if (slot < 0) return;
    const uint64_t sb = d.leaf_self[g], ob = d.leaf_opp[g];
    uint32_t cs = cache_slot(d, sb, ob);
    int own = 0;
    if (lane == 0) {
        // statistics: first evaluation of this position since the clear, or a repeat?  (second hash: the slot hash's
        // finaliser run once more over the swapped pair; 64 bits per entry, so false "seen" stays below ~1 %)
        uint64_t h = sb * 0x9E3779B97F4A7C15ULL ^ (ob + 0xD1B54A32D192ED03ULL);
        h = (h ^ (h >> 32)) * 0xD6E8FEB86659FD93ULL;
        h = (h ^ (h >> 32)) * 0xD6E8FEB86659FD93ULL;
        h ^= h >> 32;
        const uint64_t bit = h & (((uint64_t)d.cmask + 1) * 64 - 1);
        const unsigned long long m = 1ULL << (bit & 63);
        const bool seen = atomicOr(&d.cseen[bit >> 6], m) & m;
        unsigned long long* st = d.cstat + (size_t)blockIdx.x * 4;
        atomicAdd(&st[seen ? 1 : 0], 1ULL);
        // the way to write: the position's own entry if it is there already (evaluated twice since the clear: refresh it), an
        // empty way, else the way inserted longer ago; if another wave claimed that (empty / older) way in THIS launch, the
        // other way; if that is taken too, the result is not cached (it was evaluated anyway)
        const ulonglong2 ka = *(const ulonglong2*)(d.ck + 2 * (size_t)cs), kb = *(const ulonglong2*)(d.ck + 2 * (size_t)cs + 2);
        const bool ea = ka.x == ~0ULL && ka.y == ~0ULL, eb = kb.x == ~0ULL && kb.y == ~0ULL;
        int first;
        bool by_key = true;
        if (ka.x == sb && ka.y == ob) first = 0;
        else if (kb.x == sb && kb.y == ob) first = 1;
        else {
            by_key = false;
            if (ea || eb) first = ea ? 0 : 1;
            else first = d.clk[cs] <= d.clk[cs + 1] ? 0 : 1;
        }
        int way = first;
        own = atomicMax(&d.clk[cs + way], d.cepoch) < d.cepoch;  // first claimant of this entry in this launch
        // the position's OWN entry already claimed in this launch (another wave of the same position): nothing to do -- the other
        // way belongs to another position and must not be evicted for a duplicate of this key
        if (!own && !by_key) {
            way = first ^ 1;
            own = atomicMax(&d.clk[cs + way], d.cepoch) < d.cepoch;
        }
        if (own) {
            const ulonglong2 kw = way == 0 ? ka : kb;
            if (!(kw.x == ~0ULL && kw.y == ~0ULL) && !(kw.x == sb && kw.y == ob)) atomicAdd(&st[2], 1ULL);   // a live entry of another position goes
            cs += (uint32_t)way;
        }
    }
    if (!__shfl(own, 0)) return;
    cs = __shfl(cs, 0);
    float* dst = d.cv + (size_t)cs * 66;
    if (lane < CELLS) dst[lane] = policy[(size_t)slot * NP + lane];
    if (lane == 0) {
        dst[CELLS] = policy[(size_t)slot * NP + CELLS];
        dst[65] = value[slot];
        d.ck[2 * (size_t)cs] = sb;
        d.ck[2 * (size_t)cs + 1] = ob;
    }
}

// Philox4x32-10 keyed by (seed), counter (game id, ply, 0x2545F491, 0x9E3779B9): one uniform double in [0,1)
// per decision (restated in oracle/othello_oracle.c orc_philox_uniform; the two are compared bit for bit)
__device__ inline double philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t x0 = c0, x1 = c1, x2 = 0x2545F491u, x3 = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
        const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // 53 random bits, the construction numpy's random_sample uses: (a>>5, b>>6)
    return ((double)(x0 >> 5) * 67108864.0 + (double)(x1 >> 6)) / 9007199254740992.0;
}

// (re)start the search of slot g from position (sb, ob): empty tree, root written to batch slot `slot`
__device__ __forceinline__ void begin_root(const Dev& d, int g, uint64_t sb, uint64_t ob, uint64_t lg, int slot,
                                           int lane) {
    if (slot >= 0) write_eval(d, slot, sb, ob, lg, lane);
    if (lane == 0) {
        d.n_nodes[g] = 0;
        d.n_edges[g] = 0;
        d.leaf_self[g] = sb; d.leaf_opp[g] = ob; d.leaf_legal[g] = lg;
        d.path_len[g] = 0;
        d.eval_slot[g] = slot;
        d.pend[g] = PEND_ROOT;
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_search_begin(Dev d, const uint64_t* __restrict__ sb,
                                                      const uint64_t* __restrict__ ob, int n) {
    __shared__ BlockTally bt;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d.n_eval_next = 0;
    const bool live = g < n;
    uint64_t s0 = 0, o0 = 0;
    bool cached = false;
    if (live) { s0 = sb[g]; o0 = ob[g]; cached = cache_fetch(d, g, s0, o0, lane); }
    int slot = block_alloc_eval(d, bt, live && !cached, 0, 0, 0, 0, cached ? 1 : 0);
    if (cached) slot = kCachedSlot;
    if (g >= d.n_slots) return;
    if (!live) {
        if (lane == 0) { d.g_active[g] = 0; d.pend[g] = PEND_NONE; }
        return;
    }
    if (lane == 0) { d.g_active[g] = 1; d.g_self[g] = s0; d.g_opp[g] = o0; }
    begin_root(d, g, s0, o0, legal_moves_n<BS>(s0, o0), slot, lane);
}

// all slots idle; slot g < n_start joins (starts game id g) at round g * stagger / n_start (stagger 0: at once)
__global__ void k_slots_init(Dev d, int n_start, int stagger) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    d.g_active[g] = 0;
    d.g_id[g] = -1;
    d.pend[g] = PEND_NONE;
    d.g_join[g] = g < n_start ? (int)(((long long)g * stagger) / n_start) : -1;
}

// ---- the tree kernel ------------------------------------------------------------------------------
// Phase A (every mode): expand_pending for games whose pending position has been evaluated.
// Phase B: TREE_SELECT  one descent (node.py:91-126, parallel_self_play.py:172-197): the leaf is backed up at
//                       once when terminal (mcts.py:127-130), else queued for the network;
//          TREE_PLY     the ply step (node.py:147-182, parallel_self_play.py:374-397, self_play.py:101-117):
//                       pi from the root visit counts, record (position, pi), choose the action (sample while
//                       ply < threshold, else first argmax; forced != nullptr: play forced[g]), play it, detect
//                       the end of the game, refill the slot, queue the next root; idle slots whose join round
//                       has come start their first game;
//          TREE_NONE    nothing (last launch of a stand-alone search).
template <int MODE, int BS>
__global__ __launch_bounds__(256) void k_tree(Dev d, const float* __restrict__ policy, const float* __restrict__ value,
                                              int is_log, const int32_t* __restrict__ forced, int refill) {
    constexpr int CELLS = Geo<BS>::CELLS;
    extern __shared__ uint32_t lds_path_all[];  // [4 waves][cap_path]: the path of the current descent
    __shared__ BlockTally bt;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d.n_eval_next = 0;
    const bool in_range = g < d.n_slots;
    {
        const int pend = in_range ? d.pend[g] : PEND_NONE;
        if (pend != PEND_NONE) expand_pending<BS>(d, g, lane, pend, policy, value, is_log);
    }
    if constexpr (MODE == TREE_NONE) return;
    // the descent / the ply step below read nodes and edges other lanes of this wave have just written
    __threadfence_block();
    const bool live = in_range && d.g_active[g];

    if constexpr (MODE == TREE_SELECT) {
        volatile uint32_t* lpath = lds_path_all + (threadIdx.x >> 6) * d.cap_path;
        bool need = false, terminal = false, cached = false;
        uint64_t sb = 0, ob = 0, lg = 0;
        int depth = 0;
        Edge* edges = nullptr;
        uint32_t* path = nullptr;
        if (live) {
            Node* nodes = d.nodes + (size_t)g * d.cap_nodes;
            edges = d.edges + (size_t)g * d.cap_edges;
            path = d.path + (size_t)g * d.cap_path;
            int node = 0, pv = 0;
            sb = nodes[0].self_b;
            ob = nodes[0].opp_b;
            for (;;) {
                const Node nd = nodes[node];
                const bool pass = nd.legal == 0;
                const bool act = pass ? (lane == 0) : ((nd.legal >> lane) & 1ULL);
                const int rank = __popcll(nd.legal & ((1ULL << lane) - 1ULL));
                const int ei = (int)nd.edge_base + rank;
                double score = -INFINITY;
                int e_n = 0, e_child = 0;
                if (act) {
                    const Edge e = edges[ei];
                    e_n = e.n;
                    e_child = e.child;
                    const double q = e.n == 0 ? 0.0 : e.w / (double)e.n;          // node.py:51-60
                    const float cp_p = d.c_puct * e.prior;                       // float32 product (weak python scalar)
                    const double u = (double)cp_p * d.sqrt_tab[pv] / (double)(1 + e.n);  // node.py:116
                    score = q + u;                                               // node.py:119
                }
                const double best = wave_max_f64(score);
                const unsigned long long eq = __ballot(act && score == best);
                const int L = __ffsll(eq) - 1;  // first maximum in insertion order (strict > at node.py:121)
                const int action = pass ? CELLS : L;
                const int ce = __shfl(ei, L), child = __shfl(e_child, L), nvis = __shfl(e_n, L);
                if (lane == 0) lpath[depth] = (uint32_t)ce;
                ++depth;
                apply_known_n<BS>(sb, ob, action);  // board.make_move(action), parallel_self_play.py:189
                if (child == 0) break;        // that child has no children yet: leaf
                pv = nvis;
                node = child;
            }
            lg = legal_moves_n<BS>(sb, ob);
            terminal = lg == 0 && legal_moves_n<BS>(ob, sb) == 0;  // bitboard.pyx:249-264
            if (!terminal) cached = cache_fetch(d, g, sb, ob, lane);
            need = !terminal && !cached;
        }
        int slot = block_alloc_eval(d, bt, need, live ? 1 : 0, terminal ? 1 : 0, 0, 0, cached ? 1 : 0);
        if (!live) return;
        if (cached) slot = kCachedSlot;
        if (terminal) {  // parallel_self_play.py:133-135: back up float(get_winner()) immediately
            backup_path(edges, lpath, depth, (double)winner(sb, ob), lane, 0);
            if (lane == 0) d.pend[g] = PEND_NONE;
        } else {
            if (slot >= 0) write_eval(d, slot, sb, ob, lg, lane);
            for (int i = lane; i < depth; i += 64) path[i] = lpath[i];  // for the expansion (next launch)
            if (lane == 0) {
                d.leaf_self[g] = sb; d.leaf_opp[g] = ob; d.leaf_legal[g] = lg;
                d.path_len[g] = depth;
                d.eval_slot[g] = slot;
                d.pend[g] = PEND_LEAF;
            }
        }
    } else {  // TREE_PLY
        bool next_root = false, over = false, joined = false;
        uint64_t sb = 0, ob = 0;
        int nply = 0, gid = -1;
        if (live) {
            const Node root = d.nodes[(size_t)g * d.cap_nodes];
            const Edge* edges = d.edges + (size_t)g * d.cap_edges;
            const bool pass = root.legal == 0;
            const bool act = pass ? false : ((root.legal >> lane) & 1ULL);
            const int rank = __popcll(root.legal & ((1ULL << lane) - 1ULL));
            const int n = act ? (int)edges[root.edge_base + rank].n : 0;
            const int n64 = pass ? (int)edges[root.edge_base].n : 0;
            const int total = wave_sum_i32(n) + n64;
            // counts / counts.sum() in float32 (node.py:175-177 with temperature 1)
            const float tot_f = (float)total;
            float pi = act ? (float)n / tot_f : 0.0f;
            float pi64 = pass ? (float)n64 / tot_f : 0.0f;
            // first maximum of the visit counts (np.argmax)
            const int best_n = max(wave_max_i32(act ? n : -1), pass ? n64 : -1);
            const unsigned long long eqm = __ballot(act && n == best_n);
            const int amax = pass ? CELLS : (__ffsll(eqm) - 1);
            const int ply = d.g_ply[g];
            gid = d.g_id[g];
            const bool sample = ply < d.temp_threshold;
            int action;
            if (forced) {
                action = forced[g];
            } else if (!sample) {
                action = amax;
            } else {
                // np.random.choice(NP, p=pi): cdf = cumsum(p as float64); cdf /= cdf[-1]; searchsorted(u, 'right')
                const double u = philox_uniform(d.seed, (uint32_t)gid, (uint32_t)ply);
                double c = 0.0;
                double cdf_lane = 0.0;
                for (int i = 0; i < CELLS; ++i) {  // sequential cumsum, every lane runs it identically
                    c += (double)__shfl(pi, i);
                    if (i == lane) cdf_lane = c;
                }
                const double c64 = c + (double)pi64;
                const unsigned long long gt = __ballot(lane < CELLS && cdf_lane / c64 > u);
                action = gt ? (__ffsll(gt) - 1) : CELLS;
            }
            if (d.store_late_onehot && !sample) {  // SelfPlayWorker stores the T=0 one-hot (self_play.py:87-105)
                pi = (lane == amax) ? 1.0f : 0.0f;
                pi64 = (amax == CELLS) ? 1.0f : 0.0f;
            }
            // record the sample: position bits (state planes are unpacked at compaction) and pi
            sb = d.g_self[g];
            ob = d.g_opp[g];
            const size_t hidx = (size_t)(gid & d.hist_mask);
            {
                const size_t h = hidx * kMaxPly + ply;
                float* hp = d.hist_pi + h * 65;   // rows keep the 8x8 stride; NP entries are used
                if (lane < CELLS) hp[lane] = pi;
                if (lane == 0) {
                    hp[CELLS] = pi64;
                    d.hist_bits[h * 3 + 0] = sb;
                    d.hist_bits[h * 3 + 1] = ob;
                    d.hist_bits[h * 3 + 2] = root.legal;
                }
            }
            apply_known_n<BS>(sb, ob, action);  // game.board.make_move(action)
            nply = ply + 1;
            over = is_terminal_n<BS>(sb, ob) || nply >= kMaxPly;
            if (over) {
                int new_id = -1;
                if (lane == 0) {
                    d.game_winner[hidx] = winner(sb, ob);  // relative to the side to move at the end (L16)
                    d.game_len[hidx] = nply;
                    d.done_list[atomicAdd(&d.status[ST_DONE], 1) & d.hist_mask] = gid;
                    if (refill) {
                        const int nid = atomicAdd(&d.status[ST_NEXT], 1);
                        if (nid < d.game_limit) {
                            if (d.game_len[nid & d.hist_mask] == -1) new_id = nid;
                            else atomicOr(&d.status[ST_FLAGS], FLAG_HIST_OVERFLOW);  // ring entry not harvested yet
                        }
                    }
                }
                new_id = __shfl(new_id, 0);
                if (new_id >= 0) {
                    gid = new_id;
                    nply = 0;
                    sb = Geo<BS>::start_self();
                    ob = Geo<BS>::start_opp();
                    next_root = true;
                    if (lane == 0) d.game_len[gid & d.hist_mask] = 0;
                }
            } else {
                next_root = true;
            }
        } else if (in_range && d.g_join[g] == d.round && g < d.game_limit) {  // first game of this slot: id = slot
            gid = g;
            nply = 0;
            sb = Geo<BS>::start_self();
            ob = Geo<BS>::start_opp();
            next_root = joined = true;
            if (lane == 0) {
                d.game_len[gid & d.hist_mask] = 0;
                d.g_active[g] = 1;
                atomicAdd(&d.status[ST_ACTIVE], 1);
            }
        }
        const bool cached = next_root ? cache_fetch(d, g, sb, ob, lane) : false;
        int slot = block_alloc_eval(d, bt, next_root && !cached, 0, 0, live ? 1 : 0, over ? 1 : 0, cached ? 1 : 0);
        if (!live && !joined) return;
        if (cached) slot = kCachedSlot;
        if (next_root) {
            if (lane == 0) { d.g_id[g] = gid; d.g_self[g] = sb; d.g_opp[g] = ob; d.g_ply[g] = nply; }
            begin_root(d, g, sb, ob, legal_moves_n<BS>(sb, ob), slot, lane);
        } else if (lane == 0) {
            d.g_active[g] = 0;
            d.g_id[g] = -1;
            d.pend[g] = PEND_NONE;
            atomicSub(&d.status[ST_ACTIVE], 1);
        }
    }
}

// ---- search results (node.py:147-182 + root statistics) ------------------------------------------
template <int BS>
__global__ __launch_bounds__(256) void k_results(Dev d, int n, int temp_zero, float* __restrict__ pi_out,
                                                 int32_t* __restrict__ visits, double* __restrict__ wsum,
                                                 float* __restrict__ prior) {
    constexpr int CELLS = Geo<BS>::CELLS, NP = Geo<BS>::NPOL;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n) return;
    const Node root = d.nodes[(size_t)g * d.cap_nodes];
    const Edge* edges = d.edges + (size_t)g * d.cap_edges;
    const bool pass = root.legal == 0;
    const bool act = pass ? false : ((root.legal >> lane) & 1ULL);
    const int rank = __popcll(root.legal & ((1ULL << lane) - 1ULL));
    Edge e{0.0, 0.0f, 0, 0}, e64{0.0, 0.0f, 0, 0};
    if (act) e = edges[root.edge_base + rank];
    if (pass) e64 = edges[root.edge_base];
    const int total = wave_sum_i32(act ? (int)e.n : 0) + (int)e64.n;
    float pi, pi64;
    if (temp_zero) {
        const int best_n = max(wave_max_i32(act ? (int)e.n : -1), pass ? (int)e64.n : -1);
        const unsigned long long eqm = __ballot(act && (int)e.n == best_n);
        const int amax = pass ? CELLS : (__ffsll(eqm) - 1);
        pi = lane == amax ? 1.0f : 0.0f;
        pi64 = amax == CELLS ? 1.0f : 0.0f;
    } else {
        const float tf = (float)total;
        pi = act ? (float)e.n / tf : 0.0f;
        pi64 = pass ? (float)e64.n / tf : 0.0f;
    }
    const size_t o = (size_t)g * NP;
    const bool sq = lane < CELLS;
    if (pi_out) { if (sq) pi_out[o + lane] = pi; if (lane == 0) pi_out[o + CELLS] = pi64; }
    if (visits) { if (sq) visits[o + lane] = act ? (int)e.n : 0; if (lane == 0) visits[o + CELLS] = (int)e64.n; }
    if (wsum) { if (sq) wsum[o + lane] = act ? e.w : 0.0; if (lane == 0) wsum[o + CELLS] = e64.w; }
    if (prior) { if (sq) prior[o + lane] = act ? e.prior : 0.0f; if (lane == 0) prior[o + CELLS] = e64.prior; }
}

// ---- compaction of the replay tuples, game-major then ply (parallel_self_play.py:400-405) --------
// `list` = game ids to harvest, in output order (nullptr: ids 0..n-1); history entry of id = id & mask.
__global__ void k_scan_lengths(const int32_t* __restrict__ len, const int32_t* __restrict__ list, int mask,
                               int64_t* __restrict__ off, int32_t* __restrict__ len_out, int n, int64_t* total) {
    // single block exclusive scan (n <= a few 100k): each thread scans a contiguous chunk
    __shared__ int64_t part[1024];
    const int t = threadIdx.x, T = blockDim.x;
    const int per = (n + T - 1) / T;
    const int lo = t * per, hi = min(lo + per, n);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += max(0, len[(list ? list[i] : i) & mask]);
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t acc = 0;
        for (int i = 0; i < T; ++i) { const int64_t v = part[i]; part[i] = acc; acc += v; }
        *total = acc;
    }
    __syncthreads();
    int64_t acc = part[t];
    for (int i = lo; i < hi; ++i) {
        const int l = max(0, len[(list ? list[i] : i) & mask]);
        off[i] = acc;
        len_out[i] = l;
        acc += l;
    }
}

template <int BS>
__global__ __launch_bounds__(256) void k_compact(const uint64_t* __restrict__ hist_bits, const float* __restrict__ hist_pi,
                                                 const int32_t* __restrict__ len_out, const int32_t* __restrict__ win,
                                                 const int32_t* __restrict__ list, int mask,
                                                 const int64_t* __restrict__ off, int num_games,
                                                 float* __restrict__ states, float* __restrict__ pis, float* __restrict__ zs) {
    // one wave per (game, ply) sample; 768 B + 260 B + 4 B written per sample (8x8), coalesced rows
    constexpr int CELLS = Geo<BS>::CELLS, NP = Geo<BS>::NPOL;
    const int lane = threadIdx.x & 63;
    const int64_t w = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int gi = (int)(w / kMaxPly), ply = (int)(w % kMaxPly);
    if (gi >= num_games || ply >= len_out[gi]) return;
    const size_t hidx = (size_t)((list ? list[gi] : gi) & mask);
    const size_t h = hidx * kMaxPly + ply;
    const int64_t o = off[gi] + ply;
    const uint64_t sb = hist_bits[h * 3], ob = hist_bits[h * 3 + 1], lg = hist_bits[h * 3 + 2];
    float* st = states + o * 3 * CELLS;
    if (lane < CELLS) {
        st[lane] = (sb >> lane) & 1ULL ? 1.0f : 0.0f;        // get_tensor_input planes (bitboard.pyx:309-323)
        st[CELLS + lane] = (ob >> lane) & 1ULL ? 1.0f : 0.0f;
        st[2 * CELLS + lane] = (lg >> lane) & 1ULL ? 1.0f : 0.0f;
        pis[o * NP + lane] = hist_pi[h * 65 + lane];
    }
    if (lane == 0) {
        pis[o * NP + CELLS] = hist_pi[h * 65 + CELLS];
        const int player = (ply & 1) ? -1 : 1;           // parallel_self_play.py:385
        zs[o] = (float)(win[hidx] * player);              // parallel_self_play.py:404
    }
}

// policy_probs = torch.exp(policy_logits) (mcts.py:189, parallel_self_play.py:72-76) with the engine's own expf
__global__ void k_policy_exp(const float* __restrict__ logp, float* __restrict__ probs, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        probs[i] = expf(logp[i]);
}

// harvested history entries become free again (streaming mode)
__global__ void k_release(int32_t* __restrict__ game_len, const int32_t* __restrict__ list, int mask, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) game_len[list[i] & mask] = -1;
}

}  // namespace oth

using namespace oth;

// =================================================================================================
// host side
// =================================================================================================
struct oth_engine {
    oth_engine_cfg cfg{};
    oth_net* net = nullptr;
    Dev d{};
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;   // parallel to allocs (oth_engine_snapshot copies some of them)
    int32_t* n_eval2 = nullptr;   // the two evaluation-batch cursors
    int ev_par = 0;               // cursor the NEXT tree launch fills
    // history ring + per-harvest buffers
    int32_t hist_cap = 0;
    void* hist_allocs[8] = {nullptr};
    int64_t* d_off = nullptr;
    int32_t *d_len_out = nullptr, *d_list = nullptr;
    float *out_states = nullptr, *out_pis = nullptr, *out_zs = nullptr;
    int64_t out_cap = 0, n_samples = 0;
    int64_t* d_total = nullptr;
    int device = 0;   // HIP device of every allocation of this engine
    int board = 8;    // board size: 8 (the reference's game) or 6
    int cells() const { return board * board; }
    int npol() const { return board * board + 1; }
    int32_t n_roots = 0;
    int32_t run_games = 0;        // games in the last harvest
    std::vector<int32_t> run_ids; // their ids, in output order
    bool lockstep = false, streaming = false;
    int32_t round = 0;            // ply rounds launched since the run / stream began
    int32_t harvested = 0;        // entries of done_list already harvested
    int64_t counters[8] = {0};
    // lagged status polling: the status words of round r are copied to pinned memory and checked while round r+1 runs
    int32_t* h_status = nullptr;  // pinned [2][4]
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, int>> ev_spans;  // (index of start event, kind 0=net 1=tree)
    std::vector<double> net_spans;  // (start, end) ms of every network launch of the last run, on the device's reference-event axis
    double net_ms = 0, tree_ms = 0;
    int64_t net_launches = 0, tree_launches = 0;
    // oth_engine_snapshot / oth_engine_restore: device copies of the small per-slot / bookkeeping arrays + the host fields
    struct Snap {
        bool valid = false;
        std::vector<std::pair<void*, size_t>> regions;   // (source, bytes), fixed at the first snapshot of a ring size
        std::vector<void*> copies;
        int32_t hist_cap = 0;
        int ev_par = 0, round = 0, harvested = 0, cepoch = 0, n_roots = 0;
        bool lockstep = false, streaming = false;
        int64_t counters[8] = {0};
    } snap;
    // result scratch (allocated once): pi/prior f32 [G,65], visits i32 [G,65], value sums f64 [G,65], actions i32 [G]
    float *r_pi = nullptr, *r_prior = nullptr;
    int32_t *r_visits = nullptr, *r_act = nullptr;
    double* r_wsum = nullptr;
};

static inline int blocks_for(int n_slots) { return (n_slots + 3) / 4; }

// run a statement with BS = the engine's board size as a compile-time constant (kernels are templates on it)
#define OTH_DISPATCH_BS(e, ...)              \
    do {                                     \
        if ((e)->board == 6) {               \
            constexpr int BS = 6;            \
            __VA_ARGS__;                     \
        } else {                             \
            constexpr int BS = 8;            \
            __VA_ARGS__;                     \
        }                                    \
    } while (0)

template <typename T>
static int dev_alloc(oth_engine* e, T** p, size_t count) {
    void* q = nullptr;
    OTH_HIP(hipMalloc(&q, count * sizeof(T)));
    OTH_HIP(hipMemset(q, 0, count * sizeof(T)));
    e->allocs.push_back(q);
    e->alloc_bytes.push_back(count * sizeof(T));
    *p = (T*)q;
    return OTH_OK;
}


// Process-wide reference event: spans of different engines / streams are reported on one time axis so that a
// caller can take the union of overlapping network launches (several engines on several streams).
static hipEvent_t g_ref_events[64] = {};   // one per device; created on first use under g_ref_mu
static std::mutex g_ref_mu;
static int ensure_ref_event(int dev, hipEvent_t* out) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    hipEvent_t& ev = g_ref_events[dev & 63];
    if (!ev) {
        OTH_HIP(hipEventCreate(&ev));
        OTH_HIP(hipEventRecord(ev, nullptr));
        OTH_HIP(hipEventSynchronize(ev));
    }
    if (out) *out = ev;
    return OTH_OK;
}

static int span_begin(oth_engine* e, hipStream_t s, int kind) {
    if (!e->timing) return OTH_OK;
    if (e->ev_used + 2 > e->ev_pool.size()) {
        size_t old = e->ev_pool.size(), nw = old ? old * 2 : 4096;
        e->ev_pool.resize(nw);
        for (size_t i = old; i < nw; ++i) OTH_HIP(hipEventCreate(&e->ev_pool[i]));
    }
    e->ev_spans.push_back({e->ev_used, kind});
    OTH_HIP(hipEventRecord(e->ev_pool[e->ev_used], s));
    return OTH_OK;
}
static int span_end(oth_engine* e, hipStream_t s) {
    if (!e->timing) return OTH_OK;
    OTH_HIP(hipEventRecord(e->ev_pool[e->ev_used + 1], s));
    e->ev_used += 2;
    return OTH_OK;
}
static int spans_collect(oth_engine* e) {
    if (!e->timing) return OTH_OK;
    hipEvent_t ref = nullptr;
    int rc = ensure_ref_event(e->device, &ref);
    if (rc) return rc;
    for (auto& sp : e->ev_spans) {
        float ms = 0;
        OTH_HIP(hipEventElapsedTime(&ms, e->ev_pool[sp.first], e->ev_pool[sp.first + 1]));
        if (sp.second == 0) {
            e->net_ms += ms; e->net_launches++;
            float t0 = 0;
            OTH_HIP(hipEventElapsedTime(&t0, ref, e->ev_pool[sp.first]));
            e->net_spans.push_back((double)t0);
            e->net_spans.push_back((double)t0 + (double)ms);
        } else { e->tree_ms += ms; e->tree_launches++; }
    }
    e->ev_spans.clear();
    e->ev_used = 0;
    return OTH_OK;
}
static void spans_reset(oth_engine* e) {
    e->net_spans.clear();
    e->net_ms = e->tree_ms = 0;
    e->net_launches = e->tree_launches = 0;
    e->ev_spans.clear();
    e->ev_used = 0;
}

// ---- launches -------------------------------------------------------------------------------------
// cursor filled by the most recent tree / begin launch (what the network launch that follows must read)
static inline int32_t* filled_cursor(oth_engine* e) { return e->n_eval2 + (e->ev_par ^ 1); }

static int launch_net(oth_engine* e, hipStream_t s) {
    OTH_CHECK(e->net, "engine has no network: call oth_engine_set_net (or drive an external evaluator "
                      "through oth_search_leaves/oth_search_expand)");
    int r = span_begin(e, s, 0);
    if (r) return r;
    r = oth_net_forward_bits(e->net, e->d.ev_self, e->d.ev_opp, e->d.ev_legal, e->d.n_slots, filled_cursor(e), e->d.logp,
                             e->d.val, s);
    if (r) return r;
    e->counters[4]++;
    if ((r = span_end(e, s))) return r;
    if (e->d.cmask) {  // insert the fresh results before the next tree launch looks the cache up
        e->d.cepoch += 1;
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_cache_insert<BS>), dim3(blocks_for(e->d.n_slots)), dim3(256), 0, s, e->d,
                                              e->d.logp, e->d.val));
        OTH_HIP(hipGetLastError());
    }
    return OTH_OK;
}
// bind the cursor pair for the next tree-side launch and flip it
static inline void bind_cursor(oth_engine* e) {
    e->d.n_eval = e->n_eval2 + e->ev_par;
    e->d.n_eval_next = e->n_eval2 + (e->ev_par ^ 1);
    e->ev_par ^= 1;
}
static int launch_tree(oth_engine* e, int mode, int is_log, const int32_t* forced, int refill, hipStream_t s) {
    int r = span_begin(e, s, 1);
    if (r) return r;
    bind_cursor(e);
    const dim3 grid(blocks_for(e->d.n_slots)), block(256);
    const size_t lds = sizeof(uint32_t) * 4 * e->d.cap_path;
    if (mode == TREE_SELECT)
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_tree<TREE_SELECT, BS>), grid, block, lds, s, e->d, e->d.logp, e->d.val,
                                              is_log, forced, refill));
    else if (mode == TREE_PLY)
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_tree<TREE_PLY, BS>), grid, block, 0, s, e->d, e->d.logp, e->d.val, is_log,
                                              forced, refill));
    else
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_tree<TREE_NONE, BS>), grid, block, 0, s, e->d, e->d.logp, e->d.val, is_log,
                                              forced, refill));
    OTH_HIP(hipGetLastError());
    return span_end(e, s);
}
static int cache_clear(oth_engine* e, hipStream_t s) {
    if (!e->d.cmask) return OTH_OK;
    const size_t C = (size_t)e->d.cmask + 1;
    OTH_HIP(hipMemsetAsync(e->d.ck, 0xFF, C * 2 * sizeof(uint64_t), s));
    OTH_HIP(hipMemsetAsync(e->d.clk, 0, C * sizeof(int32_t), s));
    OTH_HIP(hipMemsetAsync(e->d.cseen, 0, C * sizeof(unsigned long long), s));   // (the statistics rows are cumulative)
    e->d.cepoch = 0;
    return OTH_OK;
}
static int reset_cursors(oth_engine* e, hipStream_t s) {
    OTH_HIP(hipMemsetAsync(e->n_eval2, 0, 2 * sizeof(int32_t), s));
    e->ev_par = 0;
    return OTH_OK;
}
// roots already queued: evaluate them, then num_simulations x (expand + select, evaluate); the last expansion is
// done by the caller's closing tree launch (TREE_NONE for a stand-alone search, TREE_PLY in self-play)
static int run_search(oth_engine* e, hipStream_t s) {
    int r;
    if ((r = launch_net(e, s))) return r;
    for (int i = 0; i < e->cfg.num_simulations; ++i) {
        if ((r = launch_tree(e, TREE_SELECT, 1, nullptr, 0, s))) return r;
        if ((r = launch_net(e, s))) return r;
    }
    return OTH_OK;
}

// per-block event counters -> engine totals ([4] = network batches is host-side)
static int read_counters(oth_engine* e, hipStream_t s) {
    const size_t nb = (size_t)blocks_for(e->d.n_slots);
    std::vector<unsigned long long> hc(nb * 8);
    OTH_HIP(hipMemcpyAsync(hc.data(), e->d.counters, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 8; ++i) {
        if (i == 4) continue;
        unsigned long long t = 0;
        for (size_t b = 0; b < nb; ++b) t += hc[b * 8 + i];
        e->counters[i] = (int64_t)t;
    }
    return OTH_OK;
}

// history ring of `cap` games (power of two).  A larger ring replaces the old one; the old buffers are freed.
static int ensure_history(oth_engine* e, int min_games) {
    int cap = 64;
    while (cap < min_games) cap <<= 1;
    if (cap <= e->hist_cap) return OTH_OK;
    for (void*& p : e->hist_allocs) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    e->hist_cap = 0;
    auto grab = [&](int i, size_t bytes) -> int {
        OTH_HIP(hipMalloc(&e->hist_allocs[i], bytes));
        return OTH_OK;
    };
    int r = 0;
    r |= grab(0, sizeof(uint64_t) * (size_t)cap * kMaxPly * 3);
    r |= grab(1, sizeof(float) * (size_t)cap * kMaxPly * 65);
    r |= grab(2, sizeof(int32_t) * (size_t)cap);
    r |= grab(3, sizeof(int32_t) * (size_t)cap);
    r |= grab(4, sizeof(int32_t) * (size_t)cap);
    r |= grab(5, sizeof(int64_t) * (size_t)cap);
    r |= grab(6, sizeof(int32_t) * (size_t)cap);
    r |= grab(7, sizeof(int32_t) * (size_t)cap);
    if (r) return OTH_E_HIP;
    e->d.hist_bits = (uint64_t*)e->hist_allocs[0];
    e->d.hist_pi = (float*)e->hist_allocs[1];
    e->d.game_len = (int32_t*)e->hist_allocs[2];
    e->d.game_winner = (int32_t*)e->hist_allocs[3];
    e->d.done_list = (int32_t*)e->hist_allocs[4];
    e->d_off = (int64_t*)e->hist_allocs[5];
    e->d_len_out = (int32_t*)e->hist_allocs[6];
    e->d_list = (int32_t*)e->hist_allocs[7];
    e->hist_cap = cap;
    e->d.hist_mask = cap - 1;
    return OTH_OK;
}

// Compact the replay tuples of `n` games into the output arrays.  ids == nullptr: games 0..n-1 (a finished batch
// or lock-step run); else the given ids (ascending), whose ring entries are released afterwards.
static int harvest(oth_engine* e, int n, const int32_t* ids, int64_t* n_samples, hipStream_t s) {
    int64_t total = 0;
    const int32_t* dlist = nullptr;
    if (n > 0) {
        if (ids) {
            OTH_HIP(hipMemcpyAsync(e->d_list, ids, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
            dlist = e->d_list;
        }
        hipLaunchKernelGGL(k_scan_lengths, dim3(1), dim3(1024), 0, s, e->d.game_len, dlist, e->d.hist_mask, e->d_off,
                           e->d_len_out, n, e->d_total);
        OTH_HIP(hipGetLastError());
        OTH_HIP(hipMemcpyAsync(&total, e->d_total, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        OTH_HIP(hipStreamSynchronize(s));
    }
    if (total > e->out_cap) {
        if (e->out_states) { (void)hipFree(e->out_states); (void)hipFree(e->out_pis); (void)hipFree(e->out_zs); }
        e->out_states = e->out_pis = e->out_zs = nullptr;
        e->out_cap = 0;
        const int64_t cap = total + total / 8 + 64;
        OTH_HIP(hipMalloc(&e->out_states, (size_t)cap * 192 * sizeof(float)));
        OTH_HIP(hipMalloc(&e->out_pis, (size_t)cap * 65 * sizeof(float)));
        OTH_HIP(hipMalloc(&e->out_zs, (size_t)cap * sizeof(float)));
        e->out_cap = cap;
    }
    if (total > 0) {
        const int64_t waves = (int64_t)n * kMaxPly;
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_compact<BS>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                                              e->d.hist_bits, e->d.hist_pi, e->d_len_out, e->d.game_winner, dlist,
                                              e->d.hist_mask, e->d_off, n, e->out_states, e->out_pis, e->out_zs));
        OTH_HIP(hipGetLastError());
    }
    if (ids && n > 0) {
        hipLaunchKernelGGL(k_release, dim3((n + 255) / 256), dim3(256), 0, s, e->d.game_len, dlist, e->d.hist_mask, n);
        OTH_HIP(hipGetLastError());
    }
    int rc = read_counters(e, s);  // synchronises the stream
    if (rc) return rc;
    e->n_samples = total;
    e->run_games = n;
    if (ids) e->run_ids.assign(ids, ids + n);
    else { e->run_ids.resize((size_t)n); for (int i = 0; i < n; ++i) e->run_ids[(size_t)i] = i; }
    if (n_samples) *n_samples = total;
    return spans_collect(e);
}

// common start of a batch run / lock-step run / stream: history ring, counters, status words, slots
static int reset_run(oth_engine* e, int hist_games, int game_limit, int n_start, int stagger, uint64_t seed, hipStream_t s) {
    int r = ensure_history(e, hist_games);
    if (r) return r;
    OTH_HIP(hipMemsetAsync(e->d.game_len, 0xFF, sizeof(int32_t) * (size_t)e->hist_cap, s));  // -1: free
    OTH_HIP(hipMemsetAsync(e->d.counters, 0, sizeof(unsigned long long) * 8 * (size_t)blocks_for(e->d.n_slots), s));
    if (e->d.cmask) OTH_HIP(hipMemsetAsync(e->d.cstat, 0, sizeof(unsigned long long) * 4 * (size_t)blocks_for(e->d.n_slots), s));
    const int32_t st[4] = {0, 0, 0, n_start};
    OTH_HIP(hipMemcpyAsync(e->d.status, st, sizeof(st), hipMemcpyHostToDevice, s));
    OTH_HIP(hipStreamSynchronize(s));  // `st` lives on this stack frame
    if ((r = reset_cursors(e, s))) return r;
    e->d.game_limit = game_limit;
    e->d.seed = seed;
    e->d.round = 0;
    e->round = 0;
    e->harvested = 0;
    memset(e->counters, 0, sizeof(e->counters));
    spans_reset(e);
    if ((r = cache_clear(e, s))) return r;  // the trainer may have changed the weights since the last run
    hipLaunchKernelGGL(k_slots_init, dim3((e->d.n_slots + 255) / 256), dim3(256), 0, s, e->d, n_start, stagger);
    OTH_HIP(hipGetLastError());
    return launch_tree(e, TREE_PLY, 1, nullptr, 0, s);  // round 0: the slots whose join round is 0 start their games
}

// one ply round of every playing slot: search, then the ply step
static int run_round(oth_engine* e, const int32_t* forced, int refill, hipStream_t s) {
    int r = run_search(e, s);
    if (r) return r;
    e->round += 1;
    e->d.round = e->round;
    return launch_tree(e, TREE_PLY, 1, forced, refill, s);
}

// copy the status words of the round just enqueued to pinned memory (slot = round parity) and mark it with an event
static int poll_post(oth_engine* e, hipStream_t s) {
    const int k = e->round & 1;
    OTH_HIP(hipMemcpyAsync(e->h_status + 4 * k, e->d.status, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipEventRecord(e->poll_ev[k], s));
    return OTH_OK;
}
// wait for the status words of round `round` (posted with poll_post) and return them
static int poll_wait(oth_engine* e, int round, const int32_t** st) {
    const int k = round & 1;
    OTH_HIP(hipEventSynchronize(e->poll_ev[k]));
    *st = e->h_status + 4 * k;
    if ((*st)[ST_FLAGS] & FLAG_HIST_OVERFLOW) {
        set_error("self-play history ring overflow: a game was still unharvested when its ring entry came up again "
                  "(raise hist_games / harvest more often)");
        return OTH_E_STATE;
    }
    return OTH_OK;
}

extern "C" {

oth_engine* oth_engine_create(const oth_engine_cfg* cfg) {
    if (!device_ok()) {
        set_error("oth_engine_create: no gfx950 (MI355X) device available; there is no CPU fallback");
        return nullptr;
    }
    if (!cfg || cfg->max_games < 1 || cfg->num_simulations < 0 || cfg->num_simulations > 4000) {
        // the descent path of a simulation is staged in LDS: 4 waves x (S+2) x 4 B must fit 64 KiB
        set_error("oth_engine_create: need max_games >= 1 and 0 <= num_simulations <= 4000");
        return nullptr;
    }
    if (!(cfg->board_size == 0 || cfg->board_size == 8 || cfg->board_size == 6)) {
        set_error("oth_engine_create: board_size must be 8 (or 0 = 8) or 6");
        return nullptr;
    }
    oth_engine* e = new oth_engine();
    e->device = current_device();
    e->cfg = *cfg;
    e->board = cfg->board_size == 6 ? 6 : 8;
    Dev& d = e->d;
    const int G = cfg->max_games, S = cfg->num_simulations;
    d.n_slots = G;
    d.cap_nodes = S + 2;
    d.cap_edges = (S + 2) * kEdgesPerNode;
    d.cap_path = S + 2;
    d.num_sims = S;
    d.temp_threshold = cfg->temperature_threshold;
    d.store_late_onehot = cfg->store_late_onehot;
    d.c_puct = cfg->c_puct;
    int r = 0;
    r |= dev_alloc(e, &d.g_self, G); r |= dev_alloc(e, &d.g_opp, G);
    r |= dev_alloc(e, &d.g_ply, G); r |= dev_alloc(e, &d.g_id, G); r |= dev_alloc(e, &d.g_active, G);
    r |= dev_alloc(e, &d.g_join, G);
    r |= dev_alloc(e, &d.nodes, (size_t)G * d.cap_nodes);
    r |= dev_alloc(e, &d.edges, (size_t)G * d.cap_edges);
    r |= dev_alloc(e, &d.n_nodes, G); r |= dev_alloc(e, &d.n_edges, G);
    r |= dev_alloc(e, &d.path, (size_t)G * d.cap_path);
    r |= dev_alloc(e, &d.path_len, G); r |= dev_alloc(e, &d.pend, G); r |= dev_alloc(e, &d.eval_slot, G);
    r |= dev_alloc(e, &d.leaf_self, G); r |= dev_alloc(e, &d.leaf_opp, G); r |= dev_alloc(e, &d.leaf_legal, G);
    // the network kernel reads whole tiles of positions: pad the batch arrays
    r |= dev_alloc(e, &d.ev_self, (size_t)G + 64); r |= dev_alloc(e, &d.ev_opp, (size_t)G + 64);
    r |= dev_alloc(e, &d.ev_legal, (size_t)G + 64);
    r |= dev_alloc(e, &e->n_eval2, 4);
    r |= dev_alloc(e, &d.logp, ((size_t)G + 64) * 65); r |= dev_alloc(e, &d.val, (size_t)G + 64);
    r |= dev_alloc(e, &d.status, 4);
    r |= dev_alloc(e, &d.counters, (size_t)blocks_for(G) * 8);
    r |= dev_alloc(e, &e->d_total, 2);
    r |= dev_alloc(e, &e->r_pi, (size_t)G * 65); r |= dev_alloc(e, &e->r_prior, (size_t)G * 65);
    r |= dev_alloc(e, &e->r_visits, (size_t)G * 65); r |= dev_alloc(e, &e->r_wsum, (size_t)G * 65);
    r |= dev_alloc(e, &e->r_act, G);
    if (cfg->eval_cache_log2 > 0) {
        const int lg = cfg->eval_cache_log2 < 10 ? 10 : (cfg->eval_cache_log2 > 26 ? 26 : cfg->eval_cache_log2);
        const size_t C = (size_t)1 << lg;
        r |= dev_alloc(e, &d.ck, C * 2);
        r |= dev_alloc(e, &d.cv, C * 66);
        r |= dev_alloc(e, &d.clk, C);
        r |= dev_alloc(e, &d.cres, (size_t)G * 66);
        r |= dev_alloc(e, &d.cseen, C);
        r |= dev_alloc(e, &d.cstat, (size_t)blocks_for(G) * 4);
        if (!r) {
            d.cmask = (uint32_t)(C - 1);
            if (hipMemset(d.ck, 0xFF, C * 2 * sizeof(uint64_t)) != hipSuccess) r = 1;
        }
    }
    double* st = nullptr;
    r |= dev_alloc(e, &st, (size_t)S + 4);
    if (!r && hipHostMalloc((void**)&e->h_status, sizeof(int32_t) * 8) != hipSuccess) r = 1;
    if (!r && (hipEventCreateWithFlags(&e->poll_ev[0], hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&e->poll_ev[1], hipEventDisableTiming) != hipSuccess)) r = 1;
    if (r) {
        (void)hipGetLastError();
        set_error("oth_engine_create: device allocation failed");
        oth_engine_destroy(e);
        return nullptr;
    }
    e->d.n_eval = e->n_eval2;
    e->d.n_eval_next = e->n_eval2 + 1;
    std::vector<double> tab(S + 4);
    for (int i = 0; i < S + 4; ++i) tab[i] = sqrt((double)i);  // correctly rounded, == np.sqrt(int)
    if (hipMemcpy(st, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("oth_engine_create: upload failed");
        oth_engine_destroy(e);
        return nullptr;
    }
    d.sqrt_tab = st;
    return e;
}

void oth_engine_destroy(oth_engine* e) {
    if (!e) return;
    (void)bind_device(e->device);
    for (void* p : e->allocs) (void)hipFree(p);
    for (void* p : e->hist_allocs) if (p) (void)hipFree(p);
    if (e->out_states) { (void)hipFree(e->out_states); (void)hipFree(e->out_pis); (void)hipFree(e->out_zs); }
    for (void* p : e->snap.copies) if (p) (void)hipFree(p);
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto ev : e->poll_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->h_status) (void)hipHostFree(e->h_status);
    delete e;
}

int oth_engine_set_net(oth_engine* e, oth_net* net) {
    OTH_CHECK(e, "oth_engine_set_net: null engine");
    OTH_CHECK(!net || net->board == e->board, "oth_engine_set_net: the network is built for a %dx%d board, the engine for %dx%d",
              net ? net->board : 0, net ? net->board : 0, e->board, e->board);
    e->net = net;
    return OTH_OK;
}

// ---- step-wise search ----------------------------------------------------------------------------
int oth_search_begin(oth_engine* e, const uint64_t* sb, const uint64_t* ob, int32_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && sb && ob && n >= 1 && n <= e->d.n_slots, "oth_search_begin: need 1 <= n <= max_games roots");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    e->streaming = e->lockstep = false;
    // the roots are uploaded straight into the game-position arrays; k_search_begin reads them there
    OTH_HIP(hipMemcpyAsync(e->d.g_self, sb, sizeof(uint64_t) * n, hipMemcpyDefault, s));
    OTH_HIP(hipMemcpyAsync(e->d.g_opp, ob, sizeof(uint64_t) * n, hipMemcpyDefault, s));
    OTH_HIP(hipMemsetAsync(e->d.counters, 0, sizeof(unsigned long long) * 8 * (size_t)blocks_for(e->d.n_slots), s));
    memset(e->counters, 0, sizeof(e->counters));
    spans_reset(e);
    int rc = reset_cursors(e, s);
    if (rc) return rc;
    if ((rc = cache_clear(e, s))) return rc;
    bind_cursor(e);
    OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_search_begin<BS>), dim3(blocks_for(e->d.n_slots)), dim3(256), 0, s, e->d,
                                          e->d.g_self, e->d.g_opp, n));
    OTH_HIP(hipGetLastError());
    e->n_roots = n;
    return OTH_OK;
}

int oth_search_select(oth_engine* e, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->n_roots > 0, "oth_search_select: call oth_search_begin first");
    OTH_BIND(e->device);
    return launch_tree(e, TREE_SELECT, 0, nullptr, 0, as_stream(stream));  // nothing pending: descent only
}

int oth_search_leaves(oth_engine* e, int32_t* count, uint64_t* sb, uint64_t* ob, uint64_t* lg, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && count, "oth_search_leaves: null argument");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int32_t n = 0;
    OTH_HIP(hipMemcpyAsync(&n, filled_cursor(e), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    *count = n;
    if (n > 0) {
        if (sb) OTH_HIP(hipMemcpyAsync(sb, e->d.ev_self, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, s));
        if (ob) OTH_HIP(hipMemcpyAsync(ob, e->d.ev_opp, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, s));
        if (lg) OTH_HIP(hipMemcpyAsync(lg, e->d.ev_legal, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, s));
        OTH_HIP(hipStreamSynchronize(s));
    }
    return OTH_OK;
}

int oth_search_expand(oth_engine* e, const float* policy, const float* value, int32_t is_log, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && policy && value, "oth_search_expand: null argument");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int32_t n = 0;
    OTH_HIP(hipMemcpyAsync(&n, filled_cursor(e), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    if (n > 0) {  // caller memory may be host or device: stage into the engine's own arrays
        OTH_HIP(hipMemcpyAsync(e->d.logp, policy, sizeof(float) * e->npol() * n, hipMemcpyDefault, s));
        OTH_HIP(hipMemcpyAsync(e->d.val, value, sizeof(float) * n, hipMemcpyDefault, s));
        if (e->d.cmask) {
            e->d.cepoch += 1;
            OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_cache_insert<BS>), dim3(blocks_for(e->d.n_slots)), dim3(256), 0, s,
                                                  e->d, e->d.logp, e->d.val));
            OTH_HIP(hipGetLastError());
        }
    }
    return launch_tree(e, TREE_NONE, is_log ? 1 : 0, nullptr, 0, s);
}

int oth_search_run(oth_engine* e, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->n_roots > 0, "oth_search_run: call oth_search_begin first");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int r = run_search(e, s);
    if (r) return r;
    return launch_tree(e, TREE_NONE, 1, nullptr, 0, s);
}

int oth_search_results(oth_engine* e, double temperature, float* pi, int32_t* visits, double* wsum, float* prior,
                       void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->n_roots > 0, "oth_search_results: no search in progress");
    OTH_BIND(e->device);
    OTH_CHECK(temperature == 0.0 || temperature == 1.0, "oth_search_results: temperature must be 0 or 1");
    hipStream_t s = as_stream(stream);
    const int n = e->n_roots;
    float *dpi = e->r_pi, *dpr = e->r_prior;
    int32_t* dv = e->r_visits;
    double* dw = e->r_wsum;
    OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_results<BS>), dim3(blocks_for(n)), dim3(256), 0, s, e->d, n,
                                          temperature == 0.0 ? 1 : 0, dpi, dv, dw, dpr));
    OTH_HIP(hipGetLastError());
    const size_t np = (size_t)e->npol();
    if (pi) OTH_HIP(hipMemcpyAsync(pi, dpi, sizeof(float) * np * n, hipMemcpyDeviceToHost, s));
    if (visits) OTH_HIP(hipMemcpyAsync(visits, dv, sizeof(int32_t) * np * n, hipMemcpyDeviceToHost, s));
    if (wsum) OTH_HIP(hipMemcpyAsync(wsum, dw, sizeof(double) * np * n, hipMemcpyDeviceToHost, s));
    if (prior) OTH_HIP(hipMemcpyAsync(prior, dpr, sizeof(float) * np * n, hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    int rc = read_counters(e, s);
    if (rc) return rc;
    return spans_collect(e);
}

// ---- self-play -----------------------------------------------------------------------------------
int oth_selfplay_run(oth_engine* e, int32_t num_games, uint64_t seed, int32_t add_noise, int64_t* n_samples,
                     void* stream) {
    OTH_NEED_DEVICE();
    (void)add_noise;  // no observable effect on this search: see the header comment
    OTH_CHECK(e && num_games >= 1, "oth_selfplay_run: bad arguments");
    OTH_BIND(e->device);
    OTH_CHECK(e->net, "oth_selfplay_run: no network set");
    hipStream_t s = as_stream(stream);
    const int G = e->d.n_slots;
    const int n_start = num_games < G ? num_games : G;
    int r = reset_run(e, num_games, num_games, n_start, 0, seed, s);
    if (r) return r;
    e->lockstep = e->streaming = false;
    e->n_roots = 0;
    const int64_t max_rounds = (int64_t)kMaxPly * ((num_games + G - 1) / G + 1) + 2;
    // The host stays one round ahead: round r is enqueued, then the status words of round r-1 are checked, so the
    // device never waits for the host.  The round enqueued after the last game ended finds no playing slot.
    for (;;) {
        if ((r = run_round(e, nullptr, 1, s))) return r;
        if ((r = poll_post(e, s))) return r;
        if (e->round >= 2) {
            const int32_t* st = nullptr;
            if ((r = poll_wait(e, e->round - 1, &st))) return r;
            if (st[ST_ACTIVE] <= 0) break;
        }
        if (e->round > max_rounds) {
            set_error("oth_selfplay_run: games did not finish (internal error)");
            return OTH_E_STATE;
        }
    }
    {   // the round enqueued last found nothing to do; wait for it (and for its flags)
        const int32_t* st = nullptr;
        if ((r = poll_wait(e, e->round, &st))) return r;
        if (st[ST_ACTIVE] > 0) {
            set_error("oth_selfplay_run: slots still playing after the final round (internal error)");
            return OTH_E_STATE;
        }
    }
    return harvest(e, num_games, nullptr, n_samples, s);
}

int oth_selfplay_begin(oth_engine* e, int32_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && n >= 1 && n <= e->d.n_slots, "oth_selfplay_begin: need 1 <= n <= max_games");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int r = reset_run(e, n, n, n, 0, 0, s);
    if (r) return r;
    e->lockstep = true;
    e->streaming = false;
    e->n_roots = n;
    return OTH_OK;
}

int oth_selfplay_search(oth_engine* e, float* pi, int32_t* active, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->lockstep, "oth_selfplay_search: call oth_selfplay_begin first");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int r = run_search(e, s);
    if (r) return r;
    if ((r = launch_tree(e, TREE_NONE, 1, nullptr, 0, s))) return r;   // last expansion: the root statistics are final
    const int n = e->n_roots;
    if (active) OTH_HIP(hipMemcpyAsync(active, e->d.g_active, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    if (pi) {
        float* dpi = e->r_pi;
        OTH_HIP(hipMemsetAsync(dpi, 0, sizeof(float) * e->npol() * n, s));
        // finished games keep a stale tree: their rows are garbage by contract (active[i] == 0)
        OTH_DISPATCH_BS(e, hipLaunchKernelGGL((k_results<BS>), dim3(blocks_for(n)), dim3(256), 0, s, e->d, n, 0, dpi,
                                              (int32_t*)nullptr, (double*)nullptr, (float*)nullptr));
        OTH_HIP(hipGetLastError());
        OTH_HIP(hipMemcpyAsync(pi, dpi, sizeof(float) * e->npol() * n, hipMemcpyDeviceToHost, s));
    }
    OTH_HIP(hipStreamSynchronize(s));
    return OTH_OK;
}

int oth_selfplay_apply(oth_engine* e, const int32_t* actions, int32_t* n_unfinished, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->lockstep && actions, "oth_selfplay_apply: bad state or arguments");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    int32_t* dact = e->r_act;
    OTH_HIP(hipMemsetAsync(dact, 0, sizeof(int32_t) * e->d.n_slots, s));
    OTH_HIP(hipMemcpyAsync(dact, actions, sizeof(int32_t) * e->n_roots, hipMemcpyDefault, s));
    e->round += 1;
    e->d.round = e->round;
    int r = launch_tree(e, TREE_PLY, 1, dact, 0, s);  // nothing pending (oth_selfplay_search expanded everything)
    if (r) return r;
    int32_t st[4] = {0, 0, 0, 0};
    OTH_HIP(hipMemcpyAsync(st, e->d.status, sizeof(st), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    if (n_unfinished) *n_unfinished = st[ST_ACTIVE];
    return OTH_OK;
}

int oth_selfplay_end(oth_engine* e, int64_t* n_samples, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->lockstep, "oth_selfplay_end: no lock-step run in progress");
    OTH_BIND(e->device);
    e->lockstep = false;
    const int n = e->n_roots;
    e->n_roots = 0;
    return harvest(e, n, nullptr, n_samples, as_stream(stream));
}

// ---- streaming self-play -------------------------------------------------------------------------
int oth_stream_begin(oth_engine* e, uint64_t seed, int32_t stagger_rounds, int32_t hist_games, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && stagger_rounds >= 0 && hist_games >= 0, "oth_stream_begin: bad arguments");
    OTH_BIND(e->device);
    OTH_CHECK(e->net, "oth_stream_begin: no network set");
    hipStream_t s = as_stream(stream);
    const int G = e->d.n_slots;
    const int64_t want = hist_games > 0 ? hist_games : (int64_t)8 * G;
    OTH_CHECK(want >= 2 * (int64_t)G && want <= (1 << 24), "oth_stream_begin: hist_games must be in [2*max_games, 2^24]");
    int r = reset_run(e, (int)want, INT_MAX, G, stagger_rounds, seed, s);
    if (r) return r;
    e->streaming = true;
    e->lockstep = false;
    e->n_roots = 0;
    return OTH_OK;
}

int oth_stream_step(oth_engine* e, int32_t min_games, int32_t* n_games, int64_t* n_samples, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->streaming, "oth_stream_step: call oth_stream_begin first");
    OTH_CHECK(min_games >= 1 && min_games <= e->hist_cap - 2 * e->d.n_slots,
              "oth_stream_step: min_games must be in [1, hist_games - 2*max_games]");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    spans_reset(e);
    int r;
    if ((r = cache_clear(e, s))) return r;  // a step is where a trainer would have changed the weights
    // Rounds are enqueued one ahead of the check (see oth_selfplay_run): the step ends with the round that was
    // already in flight when the round before it was seen to reach the target -- a deterministic rule.
    const int first = e->round + 1;
    for (;;) {
        if ((r = run_round(e, nullptr, 1, s))) return r;
        if ((r = poll_post(e, s))) return r;
        if (e->round > first) {
            const int32_t* st = nullptr;
            if ((r = poll_wait(e, e->round - 1, &st))) return r;
            if (st[ST_DONE] - e->harvested >= min_games) break;
        }
        if (e->round - first > 4 * kMaxPly + min_games) {
            set_error("oth_stream_step: no progress (internal error)");
            return OTH_E_STATE;
        }
    }
    const int32_t* st = nullptr;
    if ((r = poll_wait(e, e->round, &st))) return r;
    const int done = st[ST_DONE];
    const int n = done - e->harvested;
    OTH_CHECK(n <= e->hist_cap, "oth_stream_step: more finished games than the history ring holds");
    std::vector<int32_t> ids((size_t)n);
    if (n > 0) {
        const int mask = e->d.hist_mask, b0 = e->harvested & mask;
        const int first_part = std::min(n, e->hist_cap - b0);
        OTH_HIP(hipMemcpyAsync(ids.data(), e->d.done_list + b0, sizeof(int32_t) * (size_t)first_part, hipMemcpyDeviceToHost, s));
        if (n > first_part)
            OTH_HIP(hipMemcpyAsync(ids.data() + first_part, e->d.done_list, sizeof(int32_t) * (size_t)(n - first_part),
                                   hipMemcpyDeviceToHost, s));
        OTH_HIP(hipStreamSynchronize(s));
        std::sort(ids.begin(), ids.end());
    }
    e->harvested = done;
    if (n_games) *n_games = n;
    return harvest(e, n, ids.data(), n_samples, s);
}

// ---- snapshot / restore (the rescue of a saturated fp16-split launch: see oth_net_saturated in the header) --------------
// At a call boundary every playing slot is in the state begin_root left it in: an empty tree (n_nodes = n_edges = 0), its
// root queued (leaf_*, eval_slot, pend = PEND_ROOT, the dense evaluation batch and its cursor) or taken from the evaluation
// cache (cres).  So the node / edge / path arenas need no copy -- the per-slot words, the batch, the status words, the
// counters and the ring's bookkeeping do (hist_bits / hist_pi are append-only per (game, ply): a replayed ply rewrites
// its own entry).
int oth_engine_snapshot(oth_engine* e, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && (e->streaming || e->lockstep), "oth_engine_snapshot: no stream or lock-step run in progress");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    auto& sn = e->snap;
    if (sn.regions.empty() || sn.hist_cap != e->hist_cap) {
        for (void* p : sn.copies) if (p) (void)hipFree(p);
        sn.copies.clear();
        sn.regions.clear();
        const Dev& d = e->d;
        const void* skip[] = {d.nodes, d.edges, d.path, d.ck, d.cv, d.clk, d.cseen, d.sqrt_tab, e->r_pi, e->r_prior, e->r_visits,
                              e->r_wsum, e->r_act, e->d_total};
        for (size_t i = 0; i < e->allocs.size(); ++i) {
            bool sk = false;
            for (const void* q : skip) sk = sk || q == e->allocs[i];
            if (!sk) sn.regions.push_back({e->allocs[i], e->alloc_bytes[i]});
        }
        const size_t hb = sizeof(int32_t) * (size_t)e->hist_cap;
        sn.regions.push_back({d.game_len, hb});
        sn.regions.push_back({d.game_winner, hb});
        sn.regions.push_back({d.done_list, hb});
        for (auto& r : sn.regions) {
            void* c = nullptr;
            OTH_HIP(hipMalloc(&c, r.second));
            sn.copies.push_back(c);
        }
        sn.hist_cap = e->hist_cap;
    }
    for (size_t i = 0; i < sn.regions.size(); ++i)
        OTH_HIP(hipMemcpyAsync(sn.copies[i], sn.regions[i].first, sn.regions[i].second, hipMemcpyDeviceToDevice, s));
    sn.ev_par = e->ev_par; sn.round = e->round; sn.harvested = e->harvested; sn.cepoch = e->d.cepoch;
    sn.n_roots = e->n_roots; sn.lockstep = e->lockstep; sn.streaming = e->streaming;
    memcpy(sn.counters, e->counters, sizeof(sn.counters));
    sn.valid = true;
    return OTH_OK;
}

int oth_engine_restore(oth_engine* e, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && e->snap.valid && e->snap.hist_cap == e->hist_cap, "oth_engine_restore: no snapshot of this run");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    auto& sn = e->snap;
    OTH_HIP(hipStreamSynchronize(s));   // the abandoned call has ended (its last poll / harvest synchronised); be sure
    for (size_t i = 0; i < sn.regions.size(); ++i)
        OTH_HIP(hipMemcpyAsync(sn.regions[i].first, sn.copies[i], sn.regions[i].second, hipMemcpyDeviceToDevice, s));
    e->ev_par = sn.ev_par; e->round = sn.round; e->d.round = sn.round; e->harvested = sn.harvested;
    e->n_roots = sn.n_roots; e->lockstep = sn.lockstep; e->streaming = sn.streaming;
    memcpy(e->counters, sn.counters, sizeof(sn.counters));
    int r = cache_clear(e, s);   // entries inserted by the abandoned call hold clamped outputs
    if (r) return r;
    e->d.cepoch = 0;
    spans_reset(e);
    OTH_HIP(hipStreamSynchronize(s));
    return OTH_OK;
}

int oth_selfplay_game_ids(oth_engine* e, int32_t* ids, int32_t capacity, int32_t* count) {
    OTH_CHECK(e && count, "oth_selfplay_game_ids: null argument");
    *count = e->run_games;
    if (ids) {
        const int m = e->run_games < capacity ? e->run_games : capacity;
        for (int i = 0; i < m; ++i) ids[i] = e->run_ids[(size_t)i];
    }
    return OTH_OK;
}

int oth_policy_exp(const float* logp, float* probs, int64_t n, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(n >= 0 && (n == 0 || (logp && probs)), "oth_policy_exp: bad arguments");
    if (n == 0) return OTH_OK;
    OTH_BIND_PTR(logp);
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_policy_exp, dim3((unsigned)g), dim3(256), 0, as_stream(stream), logp, probs, n);
    OTH_HIP(hipGetLastError());
    return OTH_OK;
}

int oth_selfplay_fetch(oth_engine* e, float* states, float* pis, float* zs, int32_t* game_len, void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e, "oth_selfplay_fetch: null engine");
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    const int64_t n = e->n_samples;
    if (n > 0) {
        if (states) OTH_HIP(hipMemcpyAsync(states, e->out_states, (size_t)n * 3 * e->cells() * sizeof(float), hipMemcpyDefault, s));
        if (pis) OTH_HIP(hipMemcpyAsync(pis, e->out_pis, (size_t)n * e->npol() * sizeof(float), hipMemcpyDefault, s));
        if (zs) OTH_HIP(hipMemcpyAsync(zs, e->out_zs, (size_t)n * sizeof(float), hipMemcpyDefault, s));
    }
    if (game_len && e->run_games > 0)
        OTH_HIP(hipMemcpyAsync(game_len, e->d_len_out, sizeof(int32_t) * e->run_games, hipMemcpyDefault, s));
    OTH_HIP(hipStreamSynchronize(s));
    return OTH_OK;
}

int oth_selfplay_device_ptrs(oth_engine* e, float** states, float** pis, float** zs, int64_t* n_samples) {
    OTH_CHECK(e, "oth_selfplay_device_ptrs: null engine");
    if (states) *states = e->out_states;
    if (pis) *pis = e->out_pis;
    if (zs) *zs = e->out_zs;
    if (n_samples) *n_samples = e->n_samples;
    return OTH_OK;
}

int oth_engine_counters(oth_engine* e, int64_t out[8]) {
    OTH_CHECK(e && out, "oth_engine_counters: null argument");
    for (int i = 0; i < 8; ++i) out[i] = e->counters[i];
    return OTH_OK;
}

int oth_engine_cache_stats(oth_engine* e, int64_t out[4], void* stream) {
    OTH_NEED_DEVICE();
    OTH_CHECK(e && out, "oth_engine_cache_stats: null argument");
    out[0] = out[1] = out[2] = 0;
    out[3] = e->d.cmask ? (int64_t)e->d.cmask + 1 : 0;
    if (!e->d.cmask) return OTH_OK;
    OTH_BIND(e->device);
    hipStream_t s = as_stream(stream);
    const size_t nb = (size_t)blocks_for(e->d.n_slots);
    std::vector<unsigned long long> h(nb * 4);
    OTH_HIP(hipMemcpyAsync(h.data(), e->d.cstat, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OTH_HIP(hipStreamSynchronize(s));
    for (size_t b = 0; b < nb; ++b)
        for (int i = 0; i < 3; ++i) out[i] += (int64_t)h[b * 4 + i];
    return OTH_OK;
}

int oth_engine_set_timing(oth_engine* e, int32_t enable) {
    OTH_CHECK(e, "oth_engine_set_timing: null engine");
    e->timing = enable != 0;
    if (e->timing) {
        OTH_BIND(e->device);
        return ensure_ref_event(e->device, nullptr);
    }
    return OTH_OK;
}

int oth_engine_net_spans(oth_engine* e, double* spans, int64_t capacity, int64_t* count) {
    OTH_CHECK(e && count, "oth_engine_net_spans: null argument");
    const int64_t n = (int64_t)e->net_spans.size() / 2;
    *count = n;
    if (spans) {
        const int64_t m = n < capacity ? n : capacity;
        for (int64_t i = 0; i < 2 * m; ++i) spans[i] = e->net_spans[(size_t)i];
    }
    return OTH_OK;
}

int oth_engine_kernel_time(oth_engine* e, double* net_ms, int64_t* net_launches, double* tree_ms, int64_t* tree_launches) {
    OTH_CHECK(e, "oth_engine_kernel_time: null engine");
    if (net_ms) *net_ms = e->net_ms;
    if (net_launches) *net_launches = e->net_launches;
    if (tree_ms) *tree_ms = e->tree_ms;
    if (tree_launches) *tree_launches = e->tree_launches;
    return OTH_OK;
}

}  // extern "C"
