/*
 * othello_oracle.c -- CPU restatement of the reference's self-play hot path.   TEST INFRASTRUCTURE.
 * See othello_oracle.h for the scope rules and parity status (PINNED by tests/golden g1..g5).
 * Citations are to /root/reference.
 */
#include "othello_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================
 * Rules -- src/cython/bitboard.pyx
 * ========================================================================================== */

/* pyx:20 DIRECTIONS, pyx:24-38 masks.  The A/H masks are applied AFTER the shift, so the col-1
 * rays die on landing in file A... exactly as the reference computes it. */
/* Board size: ORC_N = 8 is the reference (everything pinned by the goldens).  ORC_N = 6 builds the same
 * algorithm on a 6x6 grid as libothello_oracle6.so -- bit i = row*6 + col, pass action 36, the same eight
 * directions (-N, +N, -1, +1, -(N+1), -(N-1), +(N-1), +(N+1)) with the reference's post-shift masks carried over
 * (col 0 cleared where the reference clears file A, col N-1 where it clears file H; bits >= N*N never survive).
 * The reference implements NO 6x6 rules (its `game.size` is never read), so the 6x6 build has nothing to be
 * pinned against: PARITY UNPINNED.  It exists to check the HIP engine's 6x6 kernel path (BASELINE configs[4]). */
#ifndef ORC_N
#define ORC_N 8
#endif
#define ORC_CELLS (ORC_N * ORC_N)
#define ORC_NPOL (ORC_CELLS + 1)
#define ORC_PLANES (3 * ORC_CELLS)
static const int ORC_DIRS[8] = {-ORC_N, ORC_N, -1, 1, -(ORC_N + 1), -(ORC_N - 1), ORC_N - 1, ORC_N + 1};
#if ORC_N == 8
#define NOT_A 0xFEFEFEFEFEFEFEFEULL
#define NOT_H 0x7F7F7F7F7F7F7F7FULL
#define ALLB 0xFFFFFFFFFFFFFFFFULL
#elif ORC_N == 6
#define ALLB 0x0000000FFFFFFFFFULL /* 36 cells */
#define NOT_A 0x0000000FBEFBEFBEULL /* column 0 cleared in each 6-bit row: 0b111110 x 6 */
#define NOT_H 0x00000007DF7DF7DFULL /* column 5 cleared:                   0b011111 x 6 */
#else
#error "ORC_N must be 8 or 6"
#endif
static const uint64_t ORC_MASKS[8] = {ALLB, ALLB, NOT_A, NOT_H, NOT_A, NOT_H, NOT_A, NOT_H};
int orc_board_size(void) { return ORC_N; }

uint64_t orc_flip_direction(int pos, int direction, uint64_t self_b, uint64_t opp_b, uint64_t mask) {
    /* pyx:86-114: walk one ray from pos; collect opponent stones; keep them only if the first
     * non-opponent cell (after masking) is an own stone. */
    uint64_t flip = 0, cursor;
    if (direction > 0) {
        int sh = direction;
        cursor = ((1ULL << pos) << sh) & mask;
        while (cursor & opp_b) {
            flip |= cursor;
            cursor = (cursor << sh) & mask;
        }
    } else {
        int sh = -direction;
        cursor = ((1ULL << pos) >> sh) & mask;
        while (cursor & opp_b) {
            flip |= cursor;
            cursor = (cursor >> sh) & mask;
        }
    }
    if (!(cursor & self_b)) flip = 0;
    return flip;
}

uint64_t orc_flip_bits(int pos, uint64_t self_b, uint64_t opp_b) { /* pyx:127-133 */
    uint64_t f = 0;
    for (int i = 0; i < 8; ++i) f |= orc_flip_direction(pos, ORC_DIRS[i], self_b, opp_b, ORC_MASKS[i]);
    return f;
}

uint64_t orc_legal(uint64_t self_b, uint64_t opp_b) { /* pyx:148-158: brute force over empties */
    uint64_t empty = ~(self_b | opp_b) & ALLB, legal = 0;
    for (int pos = 0; pos < ORC_CELLS; ++pos)
        if ((empty >> pos) & 1ULL)
            if (orc_flip_bits(pos, self_b, opp_b) != 0) legal |= 1ULL << pos;
    return legal;
}

void orc_reset(orc_board *b) { /* pyx:62-69 */
    /* 8x8: black E4,D5 = bits 28,35; white D4,E5 = bits 27,36 (pyx:64-65); 6x6: the same centre pattern */
    const int lo = ORC_N / 2 - 1, hi = ORC_N / 2;
    b->self_board = (1ULL << (lo * ORC_N + hi)) | (1ULL << (hi * ORC_N + lo));
    b->opp_board = (1ULL << (lo * ORC_N + lo)) | (1ULL << (hi * ORC_N + hi));
    b->move_count = 0;
    b->passed = 0;
}

static void swap_players(orc_board *b) { /* pyx:160-164 */
    uint64_t t = b->self_board;
    b->self_board = b->opp_board;
    b->opp_board = t;
}

int orc_make_move(orc_board *b, int pos) { /* pyx:209-247 */
    if (pos == ORC_CELLS) {
        if (orc_legal(b->self_board, b->opp_board) == 0) {
            swap_players(b);
            b->move_count += 1;
            b->passed = 1;
            return 1;
        }
        return 0;
    }
    if (pos < 0 || pos > ORC_CELLS - 1) return 0;
    uint64_t bit = 1ULL << pos;
    if ((b->self_board | b->opp_board) & bit) return 0;
    uint64_t flip = orc_flip_bits(pos, b->self_board, b->opp_board);
    if (flip == 0) return 0;
    b->self_board |= bit | flip;
    b->opp_board &= ~flip;
    swap_players(b);
    b->move_count += 1;
    b->passed = 0;
    return 1;
}

int orc_is_terminal(const orc_board *b) { /* pyx:255-264 */
    if (orc_legal(b->self_board, b->opp_board) != 0) return 0;
    return orc_legal(b->opp_board, b->self_board) == 0;
}

int orc_popcount(uint64_t x) { /* pyx:284-290 */
    int c = 0;
    while (x) {
        c++;
        x &= x - 1;
    }
    return c;
}

int orc_winner(const orc_board *b) { /* pyx:274-282: relative to the side to move */
    int s = orc_popcount(b->self_board), o = orc_popcount(b->opp_board);
    return s > o ? 1 : (s < o ? -1 : 0);
}

int orc_legal_list(const orc_board *b, int *out) { /* pyx:177-185: ascending, or [64] */
    uint64_t legal = orc_legal(b->self_board, b->opp_board);
    if (legal == 0) {
        out[0] = ORC_CELLS;
        return 1;
    }
    int n = 0;
    for (int i = 0; i < ORC_CELLS; ++i)
        if ((legal >> i) & 1ULL) out[n++] = i;
    return n;
}

void orc_tensor(const orc_board *b, float *t) { /* pyx:309-323: own / opp / legal planes */
    uint64_t legal = orc_legal(b->self_board, b->opp_board);
    for (int i = 0; i < ORC_CELLS; ++i) {
        t[i] = ((b->self_board >> i) & 1ULL) ? 1.0f : 0.0f;
        t[ORC_CELLS + i] = ((b->opp_board >> i) & 1ULL) ? 1.0f : 0.0f;
        t[2 * ORC_CELLS + i] = ((legal >> i) & 1ULL) ? 1.0f : 0.0f;
    }
}

/* numpy.rot90(m, k) on an 8x8 (counter-clockwise) and numpy.flip(axis=-1), as used at pyx:351-368 */
static void rot90_8x8(const float *in, int k, float *out) {
    const int N = ORC_N;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) {
            int sr, sc; /* out[r][c] = in[sr][sc] */
            switch (k & 3) {
            case 0: sr = r; sc = c; break;
            case 1: sr = c; sc = N - 1 - r; break;
            case 2: sr = N - 1 - r; sc = N - 1 - c; break;
            default: sr = N - 1 - c; sc = r; break;
            }
            out[r * N + c] = in[sr * N + sc];
        }
}
static void fliplr_8x8(const float *in, float *out) {
    const int N = ORC_N;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) out[r * N + c] = in[r * N + N - 1 - c];
}

void orc_symmetries(const orc_board *b, const float *pi, float *states, float *pis) {
    /* pyx:347-370: for k in 0..3: (rot90^k), then (rot90^k followed by left-right flip); the
     * pass probability pi[ORC_CELLS] is copied unchanged. */
    float t[ORC_PLANES], tmp[ORC_CELLS];
    orc_tensor(b, t);
    for (int k = 0; k < 4; ++k) {
        float *s0 = states + (2 * k) * ORC_PLANES, *s1 = states + (2 * k + 1) * ORC_PLANES;
        float *p0 = pis + (2 * k) * ORC_NPOL, *p1 = pis + (2 * k + 1) * ORC_NPOL;
        for (int ch = 0; ch < 3; ++ch) {
            rot90_8x8(t + ch * ORC_CELLS, k, s0 + ch * ORC_CELLS);
            fliplr_8x8(s0 + ch * ORC_CELLS, s1 + ch * ORC_CELLS);
        }
        rot90_8x8(pi, k, p0);
        fliplr_8x8(p0, tmp);
        memcpy(p1, tmp, sizeof(tmp));
        p0[ORC_CELLS] = pi[ORC_CELLS];
        p1[ORC_CELLS] = pi[ORC_CELLS];
    }
}

void orc_legal_batch(const uint64_t *s, const uint64_t *o, uint64_t *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_legal(s[i], o[i]);
}
void orc_flip_batch(const uint64_t *s, const uint64_t *o, const int32_t *pos, uint64_t *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = (pos[i] >= 0 && pos[i] < ORC_CELLS) ? orc_flip_bits(pos[i], s[i], o[i]) : 0;
}

/* Same position stream and accumulators as tests/golden/make_golden.py (checksum of checksums). */
void orc_rules_checksum(int64_t n, uint64_t *legal_acc, uint64_t *flip_acc) {
    uint64_t x = 0x9E3779B97F4A7C15ULL, la = 0, fa = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t a, c, d;
        x = x * 6364136223846793005ULL + 1442695040888963407ULL; a = x;
        x = x * 6364136223846793005ULL + 1442695040888963407ULL; c = x;
        x = x * 6364136223846793005ULL + 1442695040888963407ULL; d = x;
        uint64_t occ = (i & 1) ? (a | (c & d)) : (a & c);
        uint64_t s = occ & d & ALLB, o = occ & ~d & ALLB;
        uint64_t lb = orc_legal(s, o);
        la = la * 0x100000001B3ULL + lb;
        if (lb) {
            int mv = __builtin_ctzll(lb);
            fa = fa * 0x100000001B3ULL + orc_flip_bits(mv, s, o);
        }
    }
    *legal_acc = la;
    *flip_acc = fa;
}

/* ============================================================================================
 * Search -- src/mcts/node.py, src/mcts/mcts.py, src/train/parallel_self_play.py
 * ========================================================================================== */

/* numpy's float32 add.reduce over a contiguous array (pairwise sum with 8 partial sums), which is
 * what `masked_probs.sum()` at node.py:76 evaluates.  n <= 128 here (n == ORC_NPOL). */
static float np_sum_f32(const float *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

typedef struct {
    orc_board board;   /* position at this node */
    int expanded;      /* node.py:44 is_expanded / has children */
    int n_children;
    int first_edge;    /* children are edges [first_edge, first_edge+n_children), legal order */
} t_node;

typedef struct {
    int action;        /* 0..64 */
    float prior;       /* np.float32 (node.py:84-85) */
    int visit_count;   /* node.py:38 */
    double value_sum;  /* node.py:39 (python float) */
    int child;         /* node index once that child has been expanded, else -1 */
} t_edge;

typedef struct {
    t_node *nodes;
    t_edge *edges;
    int n_nodes, n_edges, cap_nodes, cap_edges;
    double root_prior[ORC_NPOL]; /* root priors after the Dirichlet mix (float64 in the reference) */
    /* pending simulation (lock-step form) */
    int *path;
    int path_len;
    orc_board leaf_board;
    int leaf_parent_visits_dummy;
} t_tree;

static void tree_init(t_tree *t, int sims) {
    t->cap_nodes = sims + 2;
    t->cap_edges = (sims + 2) * ORC_CELLS + 8;
    t->nodes = (t_node *)malloc(sizeof(t_node) * t->cap_nodes);
    t->edges = (t_edge *)malloc(sizeof(t_edge) * t->cap_edges);
    t->path = (int *)malloc(sizeof(int) * (sims + 4));
    t->n_nodes = t->n_edges = 0;
    t->path_len = 0;
    memset(t->root_prior, 0, sizeof(t->root_prior));
}
static void tree_free(t_tree *t) {
    free(t->nodes);
    free(t->edges);
    free(t->path);
}

/* node.py:62-89 expand: masked renormalised priors over the legal actions, children in legal order */
static int tree_expand(t_tree *t, const orc_board *b, const float *probs65) {
    int legal[ORC_NPOL];
    int nl = orc_legal_list(b, legal);
    float masked[ORC_NPOL];
    memset(masked, 0, sizeof(masked));
    for (int i = 0; i < nl; ++i) masked[legal[i]] = probs65[legal[i]];
    float sum = np_sum_f32(masked, ORC_NPOL);
    if (sum > 0) {
        for (int i = 0; i < ORC_NPOL; ++i) masked[i] /= sum;
    } else {
        float u = (float)(1.0 / (double)nl); /* python float stored into a float32 array */
        for (int i = 0; i < nl; ++i) masked[legal[i]] = u;
    }
    int id = t->n_nodes++;
    t_node *nd = &t->nodes[id];
    nd->board = *b;
    nd->expanded = 1;
    nd->n_children = nl;
    nd->first_edge = t->n_edges;
    for (int i = 0; i < nl; ++i) {
        t_edge *e = &t->edges[t->n_edges++];
        e->action = legal[i];
        e->prior = masked[legal[i]];
        e->visit_count = 0;
        e->value_sum = 0.0;
        e->child = -1;
    }
    return id;
}

/* node.py:91-126 select_child.  parent_visits is the visit count OF THIS NODE (0 for the root,
 * which is never backed up: mcts.py:161-172). */
static int tree_select(const t_tree *t, int node, int parent_visits, double c_puct) {
    const t_node *nd = &t->nodes[node];
    double best = -INFINITY;
    int best_e = -1;
    double sq = sqrt((double)parent_visits); /* np.sqrt(int) -> float64 */
    float cp32 = (float)c_puct;              /* python float is a weak scalar next to np.float32 */
    for (int i = 0; i < nd->n_children; ++i) {
        const t_edge *e = &t->edges[nd->first_edge + i];
        double q = e->visit_count == 0 ? 0.0 : e->value_sum / (double)e->visit_count;
        float cp_p = cp32 * e->prior;        /* float32 product */
        double u = (double)cp_p * sq / (double)(1 + e->visit_count);
        double s = q + u;
        if (s > best) {
            best = s;
            best_e = nd->first_edge + i;
        }
    }
    return best_e;
}

/* mcts.py:114-130 / parallel_self_play.py:172-197: descend to a leaf.  Returns 1 if the leaf is
 * terminal (value written), else 0 with t->leaf_board set. */
static int tree_descend(t_tree *t, double c_puct, double *terminal_value) {
    int node = 0, pv = 0;
    orc_board b = t->nodes[0].board;
    t->path_len = 0;
    for (;;) {
        int e = tree_select(t, node, pv, c_puct);
        t->path[t->path_len++] = e;
        orc_make_move(&b, t->edges[e].action);
        if (t->edges[e].child < 0) break; /* child has no children: a leaf (node.py:47-49) */
        pv = t->edges[e].visit_count;
        node = t->edges[e].child;
    }
    t->leaf_board = b;
    if (orc_is_terminal(&b)) {
        *terminal_value = (double)orc_winner(&b);
        return 1;
    }
    return 0;
}

/* mcts.py:161-168 / parallel_self_play.py:199-204 */
static void tree_backup(t_tree *t, double value) {
    for (int i = t->path_len - 1; i >= 0; --i) {
        t_edge *e = &t->edges[t->path[i]];
        e->visit_count += 1;
        e->value_sum += value;
        value = -value;
    }
}

static void tree_finish_leaf(t_tree *t, const float *probs65, float value) {
    int id = tree_expand(t, &t->leaf_board, probs65);
    t->edges[t->path[t->path_len - 1]].child = id;
    tree_backup(t, (double)value);
}

/* ---- built-in RNG (timing runs only; parity runs plug numpy in through orc_rng) ------------- */
typedef struct {
    uint64_t s[4];
} xo_state;
static uint64_t xo_next(xo_state *st) {
    uint64_t *s = st->s, r = ((s[1] * 5) << 7 | (s[1] * 5) >> 57) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
    s[3] = (s[3] << 45) | (s[3] >> 19);
    return r;
}
static void xo_seed(xo_state *st, uint64_t seed) {
    for (int i = 0; i < 4; ++i) {
        seed += 0x9E3779B97F4A7C15ULL;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        st->s[i] = z ^ (z >> 31);
    }
}
static double xo_uniform(xo_state *st) { return (double)(xo_next(st) >> 11) * (1.0 / 9007199254740992.0); }
static double xo_normal(xo_state *st) {
    double u1 = xo_uniform(st), u2 = xo_uniform(st);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
static double xo_gamma(xo_state *st, double a) { /* Marsaglia-Tsang */
    if (a < 1.0) {
        double u = xo_uniform(st);
        if (u < 1e-300) u = 1e-300;
        return xo_gamma(st, a + 1.0) * pow(u, 1.0 / a);
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = xo_normal(st), v = 1.0 + c * x;
        if (v <= 0) continue;
        v = v * v * v;
        double u = xo_uniform(st);
        if (u < 1 - 0.0331 * x * x * x * x) return d * v;
        if (log(u) < 0.5 * x * x + d * (1 - v + log(v))) return d * v;
    }
}
static void builtin_dirichlet(void *ctx, double alpha, int n, double *out) {
    xo_state *st = (xo_state *)ctx;
    double s = 0;
    for (int i = 0; i < n; ++i) {
        out[i] = xo_gamma(st, alpha);
        s += out[i];
    }
    for (int i = 0; i < n; ++i) out[i] /= s;
}
static int builtin_choice(void *ctx, const float *pi) {
    xo_state *st = (xo_state *)ctx;
    double u = xo_uniform(st), c = 0, tot = 0;
    for (int i = 0; i < ORC_NPOL; ++i) tot += pi[i];
    for (int i = 0; i < ORC_NPOL; ++i) {
        c += pi[i] / tot;
        if (u < c) return i;
    }
    for (int i = ORC_CELLS; i >= 0; --i)
        if (pi[i] > 0) return i;
    return ORC_CELLS;
}

/* mcts.py:210-228: consumes the RNG; the mixed priors are float64 and live on the root only, where
 * the exploration term is identically zero, so selection never reads them. */
static void tree_root_noise(t_tree *t, const orc_search_cfg *cfg, const orc_rng *rng) {
    t_node *root = &t->nodes[0];
    double noise[ORC_NPOL];
    rng->dirichlet(rng->ctx, cfg->dirichlet_alpha, root->n_children, noise);
    for (int i = 0; i < root->n_children; ++i) {
        t_edge *e = &t->edges[root->first_edge + i];
        float a = (float)(1.0 - cfg->dirichlet_epsilon) * e->prior; /* python float * np.float32 */
        t->root_prior[e->action] = (double)a + cfg->dirichlet_epsilon * noise[i];
    }
}

/* node.py:147-182 get_policy_distribution */
static void tree_policy(const t_tree *t, double temperature, float *pi65) {
    const t_node *root = &t->nodes[0];
    memset(pi65, 0, sizeof(float) * ORC_NPOL);
    if (root->n_children == 0) return;
    if (temperature == 0) {
        int best = 0;
        for (int i = 1; i < root->n_children; ++i) /* np.argmax: first maximum */
            if (t->edges[root->first_edge + i].visit_count > t->edges[root->first_edge + best].visit_count)
                best = i;
        pi65[t->edges[root->first_edge + best].action] = 1.0f;
        return;
    }
    float counts[ORC_NPOL], total;
    for (int i = 0; i < root->n_children; ++i) {
        float c = (float)t->edges[root->first_edge + i].visit_count;
        /* counts ** (1.0 / temperature) on a float32 array (node.py:175): numpy keeps float32 and takes its scalar-power
         * fast paths -- exponent 2.0 is np.square, 0.5 is np.sqrt (both correctly rounded) -- and powf otherwise */
        const double ex = 1.0 / temperature;
        counts[i] = ex == 1.0 ? c : (ex == 2.0 ? c * c : (ex == 0.5 ? sqrtf(c) : powf(c, (float)ex)));
    }
    total = np_sum_f32(counts, root->n_children);
    for (int i = 0; i < root->n_children; ++i)
        pi65[t->edges[root->first_edge + i].action] = counts[i] / total;
}

static void tree_root_stats(const t_tree *t, int32_t *visits, double *wsum, double *prior) {
    const t_node *root = &t->nodes[0];
    if (visits) memset(visits, 0, sizeof(int32_t) * ORC_NPOL);
    if (wsum) memset(wsum, 0, sizeof(double) * ORC_NPOL);
    if (prior) memset(prior, 0, sizeof(double) * ORC_NPOL);
    for (int i = 0; i < root->n_children; ++i) {
        const t_edge *e = &t->edges[root->first_edge + i];
        if (visits) visits[e->action] = e->visit_count;
        if (wsum) wsum[e->action] = e->value_sum;
        if (prior) prior[e->action] = t->root_prior[e->action] != 0.0 ? t->root_prior[e->action] : (double)e->prior;
    }
}

static void default_rng(orc_rng *r, xo_state *st, uint64_t seed) {
    xo_seed(st, seed);
    r->dirichlet = builtin_dirichlet;
    r->choice = builtin_choice;
    r->ctx = st;
}

int orc_search(const orc_board *board, const orc_search_cfg *cfg, orc_eval_fn eval, void *ectx,
               const orc_rng *rng, float *pi65, int32_t *visits65, double *wsum65, double *prior65) {
    t_tree t;
    float probs[ORC_NPOL], value;
    orc_rng lr;
    xo_state st;
    if (!rng || !rng->dirichlet) {
        default_rng(&lr, &st, 12345);
        rng = &lr;
    }
    tree_init(&t, cfg->num_simulations);
    eval(ectx, 1, &board->self_board, &board->opp_board, probs, &value); /* mcts.py:74-75 */
    tree_expand(&t, board, probs);                                        /* mcts.py:76-82 */
    if (cfg->add_noise) tree_root_noise(&t, cfg, rng);                    /* mcts.py:85-86 */
    for (int s = 0; s < cfg->num_simulations; ++s) {                      /* mcts.py:89-92 */
        double tv;
        if (tree_descend(&t, cfg->c_puct, &tv)) {
            tree_backup(&t, tv);
        } else {
            eval(ectx, 1, &t.leaf_board.self_board, &t.leaf_board.opp_board, probs, &value);
            tree_finish_leaf(&t, probs, value);
        }
    }
    if (pi65) tree_policy(&t, cfg->temperature, pi65);
    tree_root_stats(&t, visits65, wsum65, prior65);
    int n = t.n_nodes;
    tree_free(&t);
    return n;
}

void orc_search_batch(const orc_board *boards, int n, const orc_search_cfg *cfg, orc_eval_fn eval,
                      void *ectx, const orc_rng *rng, float *pi, int32_t *visits) {
    if (n <= 0) return;
    orc_rng lr;
    xo_state st;
    if (!rng || !rng->dirichlet) {
        default_rng(&lr, &st, 12345);
        rng = &lr;
    }
    t_tree *trees = (t_tree *)malloc(sizeof(t_tree) * n);
    uint64_t *sb = (uint64_t *)malloc(sizeof(uint64_t) * n), *ob = (uint64_t *)malloc(sizeof(uint64_t) * n);
    float *probs = (float *)malloc(sizeof(float) * ORC_NPOL * n), *vals = (float *)malloc(sizeof(float) * n);
    int *idx = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) {
        sb[i] = boards[i].self_board;
        ob[i] = boards[i].opp_board;
    }
    eval(ectx, n, sb, ob, probs, vals); /* parallel_self_play.py:109 */
    for (int i = 0; i < n; ++i) {       /* :111-118 expand then noise, game by game */
        tree_init(&trees[i], cfg->num_simulations);
        tree_expand(&trees[i], &boards[i], probs + ORC_NPOL * i);
        if (cfg->add_noise) tree_root_noise(&trees[i], cfg, rng);
    }
    for (int s = 0; s < cfg->num_simulations; ++s) { /* :121-161 */
        int m = 0;
        for (int i = 0; i < n; ++i) {
            double tv;
            if (tree_descend(&trees[i], cfg->c_puct, &tv)) {
                tree_backup(&trees[i], tv); /* :133-135 */
            } else {
                sb[m] = trees[i].leaf_board.self_board;
                ob[m] = trees[i].leaf_board.opp_board;
                idx[m++] = i;
            }
        }
        if (m > 0) {
            eval(ectx, m, sb, ob, probs, vals); /* :144 one batched call */
            for (int j = 0; j < m; ++j) tree_finish_leaf(&trees[idx[j]], probs + ORC_NPOL * j, vals[j]);
        }
    }
    for (int i = 0; i < n; ++i) {
        if (pi) tree_policy(&trees[i], cfg->temperature, pi + ORC_NPOL * i);
        if (visits) tree_root_stats(&trees[i], visits + ORC_NPOL * i, NULL, NULL);
        tree_free(&trees[i]);
    }
    free(trees); free(sb); free(ob); free(probs); free(vals); free(idx);
}

int orc_best_action(const orc_board *board, int sims, double c_puct, orc_eval_fn eval, void *ectx) {
    /* mcts.py:271-296 */
    int legal[ORC_NPOL];
    int nl = orc_legal_list(board, legal);
    if (sims < 1) return legal[0];
    orc_search_cfg cfg = {sims, c_puct, 0.3, 0.25, 0.0, 0};
    float pi[ORC_NPOL];
    orc_search(board, &cfg, eval, ectx, NULL, pi, NULL, NULL, NULL);
    int best = legal[0];
    float bp = pi[best];
    for (int i = 0; i < nl; ++i)
        if (pi[legal[i]] > bp) {
            bp = pi[legal[i]];
            best = legal[i];
        }
    return best;
}

void orc_action_evaluations(const orc_board *board, int sims, double c_puct, orc_eval_fn eval,
                            void *ectx, int32_t *out65) {
    /* mcts.py:315-362: Q of each root child scaled to int((Q+1)*50), clipped to [0,100] */
    memset(out65, 0, sizeof(int32_t) * ORC_NPOL);
    if (sims < 1) return;
    orc_search_cfg cfg = {sims, c_puct, 0.3, 0.25, 1.0, 0};
    int32_t visits[ORC_NPOL];
    double wsum[ORC_NPOL];
    int legal[ORC_NPOL];
    int nl = orc_legal_list(board, legal);
    orc_search(board, &cfg, eval, ectx, NULL, NULL, visits, wsum, NULL);
    for (int i = 0; i < nl; ++i) {
        int a = legal[i];
        double q = visits[a] == 0 ? 0.0 : wsum[a] / (double)visits[a];
        int sc = (int)((q + 1.0) * 50.0); /* python int(): truncation toward zero */
        out65[a] = sc < 0 ? 0 : (sc > 100 ? 100 : sc);
    }
}

/* ============================================================================================
 * Self-play -- src/train/self_play.py, src/train/parallel_self_play.py
 * ========================================================================================== */
static int argmax65(const float *p) { /* np.argmax: first maximum */
    int b = 0;
    for (int i = 1; i < ORC_NPOL; ++i)
        if (p[i] > p[b]) b = i;
    return b;
}

int64_t orc_selfplay_serial(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval, void *ectx,
                            const orc_rng *rng, int64_t cap, float *states, float *pis, float *zs,
                            int32_t *moves) {
    orc_rng lr;
    xo_state st;
    if (!rng || !rng->dirichlet) {
        default_rng(&lr, &st, 777);
        rng = &lr;
    }
    int64_t n = 0;
    int *player = (int *)malloc(sizeof(int) * 256);
    for (int ep = 0; ep < num_episodes; ++ep) { /* self_play.py:156-157 */
        orc_board b;
        orc_reset(&b);
        int ply = 0;
        int64_t first = n;
        while (!orc_is_terminal(&b)) { /* self_play.py:80 */
            if (cfg->max_plies && ply >= cfg->max_plies) break;
            if (n >= cap || ply >= 256) {
                free(player);
                return -1;
            }
            double temp = ply < cfg->temperature_threshold ? 1.0 : 0.0; /* :87 */
            orc_search_cfg sc = {cfg->num_simulations, cfg->c_puct, cfg->dirichlet_alpha,
                                 cfg->dirichlet_epsilon, temp, cfg->add_noise};
            orc_tensor(&b, states + n * ORC_PLANES);                             /* :90 */
            orc_search(&b, &sc, eval, ectx, rng, pis + n * ORC_NPOL, NULL, NULL, NULL); /* :93-98 */
            player[ply] = (ply % 2 == 0) ? 1 : -1;                        /* :83 */
            int a = temp == 0 ? argmax65(pis + n * ORC_NPOL) : rng->choice(rng->ctx, pis + n * ORC_NPOL); /* :108-113 */
            if (moves) moves[n] = a;
            orc_make_move(&b, a);
            ++ply;
            ++n;
        }
        int winner = orc_winner(&b); /* :120, relative to the side to move at the end */
        for (int64_t i = first; i < n; ++i) zs[i] = (float)(winner * player[i - first]); /* :127 */
    }
    free(player);
    return n;
}

/* ---- device-RNG restatement (the engine's default mode; no reference counterpart for the generator) --------
 * The reference draws actions with numpy's global Mersenne-Twister stream (parallel_self_play.py:374-383,
 * self_play.py:108-113), which serialises the games.  The HIP engine replaces the STREAM by a counter-based
 * generator keyed by (seed, game id, ply) and keeps numpy's choice() arithmetic.  Restated here independently:
 *   orc_philox4x32_10   Philox4x32-10 as published (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
 *                       easy as 1, 2, 3", SC'11, and the Random123 library's philox.h): pinned by Random123's
 *                       known-answer vectors in tests/test_oracle_golden.py;
 *   orc_philox_uniform  key = seed (lo, hi), counter = (game id, ply, 0x2545F491, 0x9E3779B9); the double is
 *                       built from the first two output words as numpy's random_sample does from two 32-bit
 *                       draws: ((a >> 5) * 2^26 + (b >> 6)) / 2^53;
 *   orc_choice_cdf      numpy.random.choice(ORC_NPOL, p=pi) given its uniform draw u (numpy/random/mtrand.pyx, choice():
 *                       cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u, side='right')), p = pi as
 *                       float64; pinned against numpy itself in tests/test_oracle_golden.py. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t x0 = ctr[0], x1 = ctr[1], x2 = ctr[2], x3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        if (round > 0) { /* bump the key between rounds (Weyl constants) */
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        uint64_t m0 = (uint64_t)0xD2511F53u * x0, m1 = (uint64_t)0xCD9E8D57u * x2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ x1 ^ k0, n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ x3 ^ k1, n3 = (uint32_t)m0;
        x0 = n0; x1 = n1; x2 = n2; x3 = n3;
    }
    out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
}
double orc_philox_uniform(uint64_t seed, uint32_t game_id, uint32_t ply) {
    const uint32_t ctr[4] = {game_id, ply, 0x2545F491u, 0x9E3779B9u};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    return ((double)(o[0] >> 5) * 67108864.0 + (double)(o[1] >> 6)) / 9007199254740992.0;
}
int orc_choice_cdf(const float *pi65, double u) {
    double cdf[ORC_NPOL], c = 0.0;
    for (int i = 0; i < ORC_NPOL; ++i) { /* np.cumsum of the float64 copy of p: sequential adds */
        c += (double)pi65[i];
        cdf[i] = c;
    }
    const double last = cdf[ORC_CELLS];
    int idx = 0; /* searchsorted(u, side='right') = number of entries <= u */
    while (idx < ORC_NPOL && cdf[idx] / last <= u) ++idx;
    return idx;
}

static int64_t selfplay_parallel_impl(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval, void *ectx,
                                      const orc_rng *rng, int use_philox, uint64_t philox_seed, int late_onehot,
                                      int64_t cap, float *states, float *pis, float *zs, int32_t *moves,
                                      int32_t *game_len);

int64_t orc_selfplay_parallel(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval, void *ectx,
                              const orc_rng *rng, int64_t cap, float *states, float *pis, float *zs,
                              int32_t *moves) {
    return selfplay_parallel_impl(cfg, num_episodes, eval, ectx, rng, 0, 0, 0, cap, states, pis, zs, moves, NULL);
}
/* oth_selfplay_run / oth_stream_* semantics: game g (0-based id) is an independent episode played as
 * ParallelSelfPlayWorker plays it (parallel_self_play.py:324-407: search at T=1, pi = visit distribution, sample
 * while ply < threshold else argmax) with the ply-p action drawn from orc_philox_uniform(seed, g, p).  Output:
 * game-major in id order; game_len[g] (may be NULL).  late_onehot = 1: the stored pi of a ply at or after the
 * threshold is the one-hot of the arg-max, as SelfPlayWorker stores it (self_play.py:87-105: the search result at
 * temperature 0, node.py:165-170).  Games are evaluated in lock-step batches of
 * cfg->num_parallel_games only to batch the evaluator calls; no game's tuples depend on the batching. */
int64_t orc_selfplay_philox(const orc_selfplay_cfg *cfg, int num_games, uint64_t seed, int late_onehot,
                            orc_eval_fn eval, void *ectx, int64_t cap, float *states, float *pis, float *zs,
                            int32_t *moves, int32_t *game_len) {
    return selfplay_parallel_impl(cfg, num_games, eval, ectx, NULL, 1, seed, late_onehot, cap, states, pis, zs, moves,
                                  game_len);
}

static int64_t selfplay_parallel_impl(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval, void *ectx,
                                      const orc_rng *rng, int use_philox, uint64_t philox_seed, int late_onehot,
                                      int64_t cap, float *states, float *pis, float *zs, int32_t *moves,
                                      int32_t *game_len) {
    orc_rng lr;
    xo_state st;
    if (!rng || !rng->dirichlet) {
        default_rng(&lr, &st, 777);
        rng = &lr;
    }
    const int MAXP = 256;
    int64_t n = 0;
    int completed = 0;
    int G = cfg->num_parallel_games;
    while (completed < num_episodes) { /* parallel_self_play.py:300-316 */
        int bs = num_episodes - completed < G ? num_episodes - completed : G;
        orc_board *bd = (orc_board *)malloc(sizeof(orc_board) * bs);
        int *mc = (int *)calloc(bs, sizeof(int)), *fin = (int *)calloc(bs, sizeof(int)),
            *win = (int *)calloc(bs, sizeof(int));
        float *hs = (float *)malloc(sizeof(float) * ORC_PLANES * MAXP * bs);
        float *hp = (float *)malloc(sizeof(float) * ORC_NPOL * MAXP * bs);
        int *hpl = (int *)malloc(sizeof(int) * MAXP * bs), *hmv = (int *)malloc(sizeof(int) * MAXP * bs);
        orc_board *act = (orc_board *)malloc(sizeof(orc_board) * bs);
        int *aidx = (int *)malloc(sizeof(int) * bs);
        float *api = (float *)malloc(sizeof(float) * ORC_NPOL * bs);
        for (int i = 0; i < bs; ++i) orc_reset(&bd[i]);
        for (;;) { /* :347 */
            int m = 0;
            for (int i = 0; i < bs; ++i)
                if (!fin[i]) {
                    act[m] = bd[i];
                    aidx[m++] = i;
                }
            if (m == 0) break;
            orc_search_cfg sc = {cfg->num_simulations, cfg->c_puct, cfg->dirichlet_alpha,
                                 cfg->dirichlet_epsilon, 1.0, cfg->add_noise}; /* :367-372 */
            orc_search_batch(act, m, &sc, eval, ectx, rng, api, NULL);
            for (int j = 0; j < m; ++j) { /* :375-397 */
                int i = aidx[j];
                if (mc[i] >= MAXP) return -1;
                double temp = mc[i] < cfg->temperature_threshold ? 1.0 : 0.0;
                int a;
                if (temp == 0) a = argmax65(api + ORC_NPOL * j);
                else if (use_philox)
                    a = orc_choice_cdf(api + ORC_NPOL * j, orc_philox_uniform(philox_seed, (uint32_t)(completed + i), (uint32_t)mc[i]));
                else a = rng->choice(rng->ctx, api + ORC_NPOL * j);
                orc_tensor(&bd[i], hs + ((int64_t)i * MAXP + mc[i]) * ORC_PLANES);
                memcpy(hp + ((int64_t)i * MAXP + mc[i]) * ORC_NPOL, api + ORC_NPOL * j, sizeof(float) * ORC_NPOL);
                if (late_onehot && temp == 0) {
                    float *row = hp + ((int64_t)i * MAXP + mc[i]) * ORC_NPOL;
                    for (int k = 0; k < ORC_NPOL; ++k) row[k] = k == a ? 1.0f : 0.0f;
                }
                hpl[i * MAXP + mc[i]] = (mc[i] % 2 == 0) ? 1 : -1;
                hmv[i * MAXP + mc[i]] = a;
                orc_make_move(&bd[i], a);
                mc[i] += 1;
                if (orc_is_terminal(&bd[i]) || (cfg->max_plies && mc[i] >= cfg->max_plies)) {
                    fin[i] = 1;
                    win[i] = orc_winner(&bd[i]);
                }
            }
        }
        for (int i = 0; i < bs; ++i) /* :400-405 game-major */
            if (game_len) game_len[completed + i] = mc[i];
        for (int i = 0; i < bs; ++i)
            for (int p = 0; p < mc[i]; ++p) {
                if (n >= cap) return -1;
                memcpy(states + n * ORC_PLANES, hs + ((int64_t)i * MAXP + p) * ORC_PLANES, sizeof(float) * ORC_PLANES);
                memcpy(pis + n * ORC_NPOL, hp + ((int64_t)i * MAXP + p) * ORC_NPOL, sizeof(float) * ORC_NPOL);
                zs[n] = (float)(win[i] * hpl[i * MAXP + p]);
                if (moves) moves[n] = hmv[i * MAXP + p];
                ++n;
            }
        completed += bs;
        free(bd); free(mc); free(fin); free(win); free(hs); free(hp); free(hpl); free(hmv);
        free(act); free(aidx); free(api);
    }
    return n;
}

#if ORC_N == 8 /* the CPU network and the CPU-baseline drivers exist for the reference's 8x8 game only */
/* ============================================================================================
 * CPU network -- src/model/net.py (eval mode, fp32)
 * ========================================================================================== */
typedef struct {
    int cin, cout, k;     /* k = 3 or 1 */
    float *w;             /* rearranged [tap][cin][cout] */
    float *scale, *shift; /* eval-mode BN: y = x*scale + shift, scale = g/sqrt(var+eps) */
} conv_bn;

struct orc_net {
    int blocks, filters;
    conv_bn stem, *res; /* res[2*blocks] */
    conv_bn pconv, vconv;
    float *pfc_w, *pfc_b;   /* [65][128] */
    float *vfc1_w, *vfc1_b; /* [256][64] */
    float *vfc2_w, *vfc2_b; /* [1][256] */
};

int64_t orc_net_blob_floats(int B, int F) {
    int64_t n = 0;
    n += (int64_t)F * 3 * 9 + 4 * F;
    n += (int64_t)2 * B * ((int64_t)F * F * 9 + 4 * F);
    n += (int64_t)2 * F + 4 * 2 + 65 * 128 + 65;
    n += (int64_t)F + 4 * 1 + 256 * 64 + 256 + 256 + 1;
    return n;
}

static const float *load_conv_bn(conv_bn *c, int cin, int cout, int k, const float *p) {
    c->cin = cin; c->cout = cout; c->k = k;
    int taps = k * k;
    c->w = (float *)malloc(sizeof(float) * taps * cin * cout);
    for (int o = 0; o < cout; ++o) /* conv.weight [cout][cin][kh][kw] */
        for (int i = 0; i < cin; ++i)
            for (int t = 0; t < taps; ++t) c->w[((int64_t)t * cin + i) * cout + o] = p[((int64_t)o * cin + i) * taps + t];
    p += (int64_t)cout * cin * taps;
    const float *g = p, *b = p + cout, *mean = p + 2 * cout, *var = p + 3 * cout;
    c->scale = (float *)malloc(sizeof(float) * cout);
    c->shift = (float *)malloc(sizeof(float) * cout);
    for (int o = 0; o < cout; ++o) { /* BatchNorm2d eval, eps = 1e-5 (net.py:25) */
        float inv = 1.0f / sqrtf(var[o] + 1e-5f);
        c->scale[o] = g[o] * inv;
        c->shift[o] = b[o] - mean[o] * c->scale[o];
    }
    return p + 4 * cout;
}
static float *dupf(const float *p, int64_t n) {
    float *q = (float *)malloc(sizeof(float) * n);
    memcpy(q, p, sizeof(float) * n);
    return q;
}

orc_net *orc_net_create(int B, int F, const float *blob, int64_t n_floats) {
    if (n_floats != orc_net_blob_floats(B, F)) return NULL;
    orc_net *net = (orc_net *)calloc(1, sizeof(orc_net));
    net->blocks = B; net->filters = F;
    const float *p = blob;
    p = load_conv_bn(&net->stem, 3, F, 3, p);
    net->res = (conv_bn *)calloc(2 * B, sizeof(conv_bn));
    for (int i = 0; i < 2 * B; ++i) p = load_conv_bn(&net->res[i], F, F, 3, p);
    p = load_conv_bn(&net->pconv, F, 2, 1, p);
    net->pfc_w = dupf(p, 65 * 128); p += 65 * 128;
    net->pfc_b = dupf(p, 65); p += 65;
    p = load_conv_bn(&net->vconv, F, 1, 1, p);
    net->vfc1_w = dupf(p, 256 * 64); p += 256 * 64;
    net->vfc1_b = dupf(p, 256); p += 256;
    net->vfc2_w = dupf(p, 256); p += 256;
    net->vfc2_b = dupf(p, 1); p += 1;
    return net;
}
static void free_cb(conv_bn *c) { free(c->w); free(c->scale); free(c->shift); }
void orc_net_destroy(orc_net *net) {
    if (!net) return;
    free_cb(&net->stem);
    for (int i = 0; i < 2 * net->blocks; ++i) free_cb(&net->res[i]);
    free(net->res);
    free_cb(&net->pconv); free_cb(&net->vconv);
    free(net->pfc_w); free(net->pfc_b); free(net->vfc1_w); free(net->vfc1_b); free(net->vfc2_w); free(net->vfc2_b);
    free(net);
}

/* in/out: [64][C] (cell-major, channels contiguous).  out = bn(conv(in)) (+res) (relu) */
static void conv_forward(const conv_bn *c, const float *in, float *out, const float *res, int relu) {
    const int cin = c->cin, cout = c->cout;
    float acc[256];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            for (int o = 0; o < cout; ++o) acc[o] = 0.f;
            if (c->k == 3) {
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = y + dy;
                    if (yy < 0 || yy > 7) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = x + dx;
                        if (xx < 0 || xx > 7) continue;
                        const float *w = c->w + (int64_t)((dy + 1) * 3 + (dx + 1)) * cin * cout;
                        const float *a = in + (yy * 8 + xx) * cin;
                        for (int i = 0; i < cin; ++i) {
                            float av = a[i];
                            const float *wr = w + (int64_t)i * cout;
                            for (int o = 0; o < cout; ++o) acc[o] = __builtin_fmaf(av, wr[o], acc[o]);
                        }
                    }
                }
            } else {
                const float *a = in + (y * 8 + x) * cin;
                for (int i = 0; i < cin; ++i) {
                    float av = a[i];
                    const float *wr = c->w + (int64_t)i * cout;
                    for (int o = 0; o < cout; ++o) acc[o] = __builtin_fmaf(av, wr[o], acc[o]);
                }
            }
            float *op = out + (y * 8 + x) * cout;
            const float *rp = res ? res + (y * 8 + x) * cout : NULL;
            for (int o = 0; o < cout; ++o) {
                float v = acc[o] * c->scale[o] + c->shift[o];
                if (rp) v += rp[o];
                if (relu && v < 0) v = 0;
                op[o] = v;
            }
        }
}

static void net_forward_one(const orc_net *net, const float *x192, float *logp65, float *v1, float *buf) {
    const int F = net->filters;
    float *a = buf, *b = buf + 64 * F, *c = buf + 128 * F;
    float in[64 * 3];
    for (int s = 0; s < 64; ++s)
        for (int ch = 0; ch < 3; ++ch) in[s * 3 + ch] = x192[ch * 64 + s];
    conv_forward(&net->stem, in, a, NULL, 1);                 /* net.py:195 */
    for (int i = 0; i < net->blocks; ++i) {                   /* net.py:198-199, 47-61 */
        conv_forward(&net->res[2 * i], a, b, NULL, 1);
        conv_forward(&net->res[2 * i + 1], b, c, a, 1);
        float *t = a; a = c; c = t;
    }
    /* policy head net.py:83-96: flatten order is (channel, cell) */
    float ph[64 * 2], pf[128], logits[65];
    conv_forward(&net->pconv, a, ph, NULL, 1);
    for (int s = 0; s < 64; ++s) {
        pf[s] = ph[s * 2];
        pf[64 + s] = ph[s * 2 + 1];
    }
    float mx = -INFINITY;
    for (int o = 0; o < 65; ++o) {
        float acc = net->pfc_b[o];
        for (int i = 0; i < 128; ++i) acc += net->pfc_w[o * 128 + i] * pf[i];
        logits[o] = acc;
        if (acc > mx) mx = acc;
    }
    double se = 0;
    for (int o = 0; o < 65; ++o) se += exp((double)(logits[o] - mx));
    float lse = (float)log(se);
    for (int o = 0; o < 65; ++o) logp65[o] = logits[o] - mx - lse;
    /* value head net.py:119-136 */
    float vh[64], h1[256];
    conv_forward(&net->vconv, a, vh, NULL, 1);
    for (int o = 0; o < 256; ++o) {
        float acc = net->vfc1_b[o];
        for (int i = 0; i < 64; ++i) acc += net->vfc1_w[o * 64 + i] * vh[i];
        h1[o] = acc > 0 ? acc : 0;
    }
    float acc = net->vfc2_b[0];
    for (int i = 0; i < 256; ++i) acc += net->vfc2_w[i] * h1[i];
    *v1 = tanhf(acc);
}

void orc_net_forward(const orc_net *net, int n, const float *x, float *logp, float *v) {
#pragma omp parallel
    {
        float *buf = (float *)malloc(sizeof(float) * 64 * net->filters * 3);
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) net_forward_one(net, x + (int64_t)i * 192, logp + (int64_t)i * 65, v + i, buf);
        free(buf);
    }
}

void orc_net_eval(void *ctx, int n, const uint64_t *sb, const uint64_t *ob, float *probs, float *values) {
    const orc_net *net = (const orc_net *)ctx;
    float *buf = (float *)malloc(sizeof(float) * 64 * net->filters * 3);
    for (int i = 0; i < n; ++i) {
        orc_board b = {sb[i], ob[i], 0, 0};
        float x[192], logp[65];
        orc_tensor(&b, x);                 /* mcts.py:205 get_tensor_input */
        net_forward_one(net, x, logp, values + i, buf);
        for (int o = 0; o < 65; ++o) probs[i * 65 + o] = expf(logp[o]); /* mcts.py:191 torch.exp */
    }
    free(buf);
}

typedef struct {
    const orc_net *net;
    int64_t evals;
} count_ctx;
static void counting_eval(void *ctx, int n, const uint64_t *sb, const uint64_t *ob, float *probs, float *values) {
    count_ctx *c = (count_ctx *)ctx;
    c->evals += n;
    orc_net_eval((void *)c->net, n, sb, ob, probs, values);
}

int64_t orc_cpu_baseline(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int plies_per_stream,
                         uint64_t seed, int64_t *n_evals, int *threads_used) {
    int64_t total = 0, evals = 0;
    int nt = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 1) reduction(+ : total, evals)
        for (int s = 0; s < streams; ++s) {
            orc_selfplay_cfg c = *cfg;
            c.max_plies = plies_per_stream;
            xo_state st;
            orc_rng r;
            default_rng(&r, &st, seed + 1000003ULL * (uint64_t)s);
            int64_t cap = plies_per_stream + 4;
            float *S = (float *)malloc(sizeof(float) * 192 * cap), *P = (float *)malloc(sizeof(float) * 65 * cap),
                  *Z = (float *)malloc(sizeof(float) * cap);
            count_ctx cc = {net, 0};
            int64_t n = orc_selfplay_serial(&c, 1, counting_eval, &cc, &r, cap, S, P, Z, NULL);
            if (n > 0) total += n;
            evals += cc.evals;
            free(S); free(P); free(Z);
        }
    }
    if (n_evals) *n_evals = evals;
    if (threads_used) *threads_used = nt;
    return total;
}

/* Phase-uniform CPU baseline: stream s first plays (s * spread) / streams plies with uniformly random legal
 * moves (untimed by the caller's clock only in the sense that no search runs: it costs microseconds), then
 * `plies_per_stream` plies of real serial self-play (self_play.py:80-117: search, record, sample/argmax, move),
 * starting a fresh game whenever one ends.  With spread ~ one game length the sample covers openings, middle
 * games and endgames (terminal leaves included) in the proportions a whole game has. */
int64_t orc_cpu_baseline_spread(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int plies_per_stream,
                                int spread, uint64_t seed, int64_t *n_evals, int *threads_used, int64_t *games_ended) {
    int64_t total = 0, evals = 0, ended = 0;
    int nt = 1;
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 1) reduction(+ : total, evals, ended)
        for (int s = 0; s < streams; ++s) {
            xo_state st;
            orc_rng r;
            default_rng(&r, &st, seed + 1000003ULL * (uint64_t)s);
            count_ctx cc = {net, 0};
            orc_board b;
            orc_reset(&b);
            int skip = spread > 0 ? (int)(((int64_t)s * spread) / streams) : 0;
            for (int i = 0; i < skip && !orc_is_terminal(&b); ++i) {
                int mv[65];
                int n = orc_legal_list(&b, mv);
                orc_make_move(&b, mv[(int)(xo_uniform(&st) * n) % n]);
            }
            float pi[65];
            for (int done = 0; done < plies_per_stream;) {
                if (orc_is_terminal(&b)) {
                    orc_reset(&b);
                    ended += 1;
                    continue;
                }
                double temp = b.move_count < cfg->temperature_threshold ? 1.0 : 0.0;
                orc_search_cfg sc = {cfg->num_simulations, cfg->c_puct, cfg->dirichlet_alpha, cfg->dirichlet_epsilon,
                                     temp, cfg->add_noise};
                orc_search(&b, &sc, counting_eval, &cc, &r, pi, NULL, NULL, NULL);
                int a = temp == 0 ? argmax65(pi) : r.choice(r.ctx, pi);
                orc_make_move(&b, a);
                ++done;
                ++total;
            }
            evals += cc.evals;
        }
    }
    if (n_evals) *n_evals = evals;
    if (threads_used) *threads_used = nt;
    if (games_ended) *games_ended = ended;
    return total;
}
/* (round 6: orc_cpu_baseline_phased also records, per stream and ply -- rec_cap rows per stream -- the ply's PHASE (plies played
 * before it), its wall time and its network evaluations, so that bench.py can weight the plies by game phase: endgame plies are
 * cheap (many simulations end in terminal nodes) and a stream that started late finishes more of them inside the budget.)
 * Time-bounded form of the phase-uniform baseline (round 5: every hardware thread, a per-stream spread): `streams` serial
 * self-play streams on `threads` OpenMP threads (0: the runtime's default), stream s starting (s * spread) / streams random
 * plies into a game; each stream plays whole plies (self_play.py:80-117) until `budget_s` seconds have passed since the
 * parallel region began AND it has played at least `min_plies`.  plies_out[s] / secs_out[s] (either may be NULL): what
 * stream s played and how long its timed plies took (its own clock, the random prefix excluded).  Returns total plies. */
int64_t orc_cpu_baseline_phased(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int threads, int min_plies,
                                double budget_s, int spread, uint64_t seed, int32_t *plies_out, double *secs_out,
                                int64_t *n_evals, int *threads_used, int64_t *games_ended, int rec_cap, int32_t *rec_phase,
                                double *rec_secs, int32_t *rec_evals) {
    int64_t total = 0, evals = 0, ended = 0;
    int nt = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    const double t_begin = omp_get_wtime();
#else
    const double t_begin = 0.0;
#endif
#pragma omp parallel
    {
#ifdef _OPENMP
#pragma omp single
        nt = omp_get_num_threads();
#endif
#pragma omp for schedule(dynamic, 1) reduction(+ : total, evals, ended)
        for (int s = 0; s < streams; ++s) {
            xo_state st;
            orc_rng r;
            default_rng(&r, &st, seed + 1000003ULL * (uint64_t)s);
            count_ctx cc = {net, 0};
            orc_board b;
            orc_reset(&b);
            int skip = spread > 0 ? (int)(((int64_t)s * spread) / streams) : 0;
            for (int i = 0; i < skip && !orc_is_terminal(&b); ++i) {
                int mv[65];
                int n = orc_legal_list(&b, mv);
                orc_make_move(&b, mv[(int)(xo_uniform(&st) * n) % n]);
            }
            float pi[65];
            int done = 0;
#ifdef _OPENMP
            const double t0 = omp_get_wtime();
#endif
            for (;;) {
#ifdef _OPENMP
                if (done >= min_plies && omp_get_wtime() - t_begin >= budget_s) break;
#else
                if (done >= min_plies) break;
#endif
                if (orc_is_terminal(&b)) {
                    orc_reset(&b);
                    ended += 1;
                    continue;
                }
                double temp = b.move_count < cfg->temperature_threshold ? 1.0 : 0.0;
                orc_search_cfg sc = {cfg->num_simulations, cfg->c_puct, cfg->dirichlet_alpha, cfg->dirichlet_epsilon,
                                     temp, cfg->add_noise};
#ifdef _OPENMP
                const double tp = omp_get_wtime();
#endif
                const int64_t e0 = cc.evals;
                const int phase = b.move_count;
                orc_search(&b, &sc, counting_eval, &cc, &r, pi, NULL, NULL, NULL);
                int a = temp == 0 ? argmax65(pi) : r.choice(r.ctx, pi);
                orc_make_move(&b, a);
                if (rec_cap > 0 && done < rec_cap) { /* the ply's phase (plies played before it), its time and evaluations */
                    const int64_t at = (int64_t)s * rec_cap + done;
                    if (rec_phase) rec_phase[at] = phase;
                    if (rec_evals) rec_evals[at] = (int32_t)(cc.evals - e0);
#ifdef _OPENMP
                    if (rec_secs) rec_secs[at] = omp_get_wtime() - tp;
#else
                    if (rec_secs) rec_secs[at] = 0.0;
#endif
                }
                ++done;
                ++total;
            }
            if (plies_out) plies_out[s] = done;
#ifdef _OPENMP
            if (secs_out) secs_out[s] = omp_get_wtime() - t0;
#else
            if (secs_out) secs_out[s] = 0.0;
#endif
            evals += cc.evals;
        }
    }
    if (n_evals) *n_evals = evals;
    if (threads_used) *threads_used = nt;
    if (games_ended) *games_ended = ended;
    return total;
}
int64_t orc_cpu_baseline_timed(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int threads, int min_plies,
                               double budget_s, int spread, uint64_t seed, int32_t *plies_out, double *secs_out,
                               int64_t *n_evals, int *threads_used, int64_t *games_ended) {
    return orc_cpu_baseline_phased(net, cfg, streams, threads, min_plies, budget_s, spread, seed, plies_out, secs_out, n_evals,
                                   threads_used, games_ended, 0, NULL, NULL, NULL);
}
#endif /* ORC_N == 8 */
