/*
 * othello_oracle.h -- CPU restatement of the reference's self-play hot path.   TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (othello_reinforcement_learning_test_amd/) never does.  Parity status: PINNED -- every
 * function below is checked against golden vectors produced by importing the reference itself
 * (tests/golden/make_golden.py, fixtures g1..g5) and, when /root/reference is present, live
 * against oracle/_ref (the reference's Cython compiled unmodified).
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 */
#ifndef OTHELLO_ORACLE_H
#define OTHELLO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- rules: src/cython/bitboard.pyx ------------------------------------------------------- */
typedef struct {
    uint64_t self_board, opp_board; /* bitboard.pxd:25-26 */
    int32_t move_count;             /* bitboard.pxd:27 */
    int32_t passed;                 /* bitboard.pxd:28 */
} orc_board;

int orc_board_size(void); /* 8 (libothello_oracle.so, pinned) or 6 (libothello_oracle6.so, -DORC_N=6: parity unpinned) */
uint64_t orc_flip_direction(int pos, int direction, uint64_t self_b, uint64_t opp_b, uint64_t mask); /* pyx:71-114 */
uint64_t orc_flip_bits(int pos, uint64_t self_b, uint64_t opp_b);                                    /* pyx:116-133 */
uint64_t orc_legal(uint64_t self_b, uint64_t opp_b);                                                 /* pyx:135-158 */
void orc_reset(orc_board *b);                                                                        /* pyx:52-69 */
int orc_make_move(orc_board *b, int pos);                                                            /* pyx:195-247 */
int orc_is_terminal(const orc_board *b);                                                             /* pyx:249-264 */
int orc_winner(const orc_board *b);                                                                  /* pyx:266-282 */
int orc_popcount(uint64_t x);                                                                        /* pyx:284-290 */
int orc_legal_list(const orc_board *b, int *out65);                                                  /* pyx:166-185 */
void orc_tensor(const orc_board *b, float *out192);                                                  /* pyx:300-323 */
void orc_symmetries(const orc_board *b, const float *pi65, float *states8x192, float *pis8x65);      /* pyx:338-370 */

/* batch helpers for numpy-side tests */
void orc_legal_batch(const uint64_t *self_b, const uint64_t *opp_b, uint64_t *out, int64_t n);
void orc_flip_batch(const uint64_t *self_b, const uint64_t *opp_b, const int32_t *pos, uint64_t *out, int64_t n);
void orc_rules_checksum(int64_t n, uint64_t *legal_acc, uint64_t *flip_acc);

/* ---- evaluator + RNG plug-ins ---------------------------------------------------------------- */
/* probs[n*65] are PROBABILITIES (what torch.exp(policy_logits) gave the reference, mcts.py:189-191),
 * values[n] the tanh head.  */
typedef void (*orc_eval_fn)(void *ctx, int n, const uint64_t *self_b, const uint64_t *opp_b,
                            float *probs, float *values);
typedef struct {
    /* numpy.random.dirichlet([alpha]*n) (mcts.py:221) and numpy.random.choice(65, p=pi)
     * (self_play.py:113).  NULL => built-in xoshiro-based generators (timing runs only). */
    void (*dirichlet)(void *ctx, double alpha, int n, double *out);
    int (*choice)(void *ctx, const float *pi65);
    void *ctx;
} orc_rng;

/* ---- search: src/mcts/node.py, src/mcts/mcts.py, src/train/parallel_self_play.py:31-216 ------ */
typedef struct {
    int32_t num_simulations;
    double c_puct;            /* python float */
    double dirichlet_alpha;
    double dirichlet_epsilon;
    double temperature;       /* 0 or 1 are bit-pinned (node.py:165-177) */
    int32_t add_noise;
} orc_search_cfg;

/* MCTS.search (mcts.py:49-98) for one position.  Outputs may be NULL.  Returns #expanded nodes. */
int orc_search(const orc_board *board, const orc_search_cfg *cfg, orc_eval_fn eval, void *eval_ctx,
               const orc_rng *rng, float *pi65, int32_t *visits65, double *value_sum65,
               double *prior65);
/* BatchMCTS.search_batch (parallel_self_play.py:80-170): n positions in lock-step, ONE batched
 * eval call per simulation step over the non-terminal leaves. */
void orc_search_batch(const orc_board *boards, int n, const orc_search_cfg *cfg, orc_eval_fn eval,
                      void *eval_ctx, const orc_rng *rng, float *pi_nx65, int32_t *visits_nx65);
/* MCTS.get_best_action (mcts.py:257-296), get_action_evaluations (mcts.py:298-362) */
int orc_best_action(const orc_board *board, int num_simulations, double c_puct, orc_eval_fn eval,
                    void *eval_ctx);
void orc_action_evaluations(const orc_board *board, int num_simulations, double c_puct,
                            orc_eval_fn eval, void *eval_ctx, int32_t *out65);

/* ---- self-play: src/train/self_play.py:52-163, parallel_self_play.py:282-407 ----------------- */
typedef struct {
    int32_t num_simulations, temperature_threshold, num_parallel_games;
    double c_puct, dirichlet_alpha, dirichlet_epsilon;
    int32_t add_noise;
    int32_t max_plies; /* stop after this many plies per game (0 = play to the end); timing only */
} orc_selfplay_cfg;

/* Outputs: states[cap*192], pis[cap*65], zs[cap], moves[cap] (may be NULL).  Returns #samples, or
 * -1 if cap was too small.  serial = SelfPlayWorker.execute_episodes; parallel =
 * ParallelSelfPlayWorker.execute_episodes (sample order: batch by batch, game-major inside). */
int64_t orc_selfplay_serial(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval,
                            void *eval_ctx, const orc_rng *rng, int64_t cap, float *states,
                            float *pis, float *zs, int32_t *moves);
int64_t orc_selfplay_parallel(const orc_selfplay_cfg *cfg, int num_episodes, orc_eval_fn eval,
                              void *eval_ctx, const orc_rng *rng, int64_t cap, float *states,
                              float *pis, float *zs, int32_t *moves);

/* ---- device-RNG self-play (the HIP engine's default mode) -------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11 / Random123 philox.h; pinned by Random123's known-answer vectors),
 * the engine's keying (seed; game id, ply) and numpy's choice() arithmetic given the uniform draw (pinned
 * against numpy.random.RandomState.choice).  orc_selfplay_philox: every game id an independent
 * ParallelSelfPlayWorker-style episode (parallel_self_play.py:324-407), tuples game-major in id order. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_philox_uniform(uint64_t seed, uint32_t game_id, uint32_t ply);
int orc_choice_cdf(const float *pi65, double u);
int64_t orc_selfplay_philox(const orc_selfplay_cfg *cfg, int num_games, uint64_t seed, int late_onehot,
                            orc_eval_fn eval, void *eval_ctx, int64_t cap, float *states, float *pis, float *zs,
                            int32_t *moves, int32_t *game_len);

/* ---- CPU network: src/model/net.py:139-205 (eval mode, fp32) -------------------------------- */
typedef struct orc_net orc_net;
/* blob = state_dict tensors in registration order, float32, int64 num_batches_tracked skipped:
 * per Conv+BN pair: conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var. */
int64_t orc_net_blob_floats(int blocks, int filters);
orc_net *orc_net_create(int blocks, int filters, const float *blob, int64_t n_floats);
void orc_net_destroy(orc_net *net);
void orc_net_forward(const orc_net *net, int n, const float *x_nx192, float *logp_nx65, float *v_n);
/* orc_eval_fn adapter: ctx = orc_net*; probs = expf(logp) */
void orc_net_eval(void *ctx, int n, const uint64_t *self_b, const uint64_t *opp_b, float *probs,
                  float *values);

/* CPU-baseline driver: `streams` independent serial self-play streams on OpenMP threads, each
 * playing `plies_per_stream` plies (bounded sample).  Returns total plies played; evals counted. */
int64_t orc_cpu_baseline(const orc_net *net, const orc_selfplay_cfg *cfg, int streams,
                         int plies_per_stream, uint64_t seed, int64_t *n_evals, int *threads_used);

/* same, phase-uniform: stream s starts (s * spread) / streams random plies into a game, so the sample covers
 * openings, middle games and endgames like a whole game does; a stream starts a new game when one ends. */
int64_t orc_cpu_baseline_spread(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int plies_per_stream,
                                int spread, uint64_t seed, int64_t *n_evals, int *threads_used, int64_t *games_ended);

/* time-bounded form with a per-stream report (bench.py's cpu_baseline since round 5): see othello_oracle.c */
int64_t orc_cpu_baseline_phased(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int threads, int min_plies,
                                double budget_s, int spread, uint64_t seed, int32_t *plies_out, double *secs_out,
                                int64_t *n_evals, int *threads_used, int64_t *games_ended, int rec_cap, int32_t *rec_phase,
                                double *rec_secs, int32_t *rec_evals);
int64_t orc_cpu_baseline_timed(const orc_net *net, const orc_selfplay_cfg *cfg, int streams, int threads, int min_plies,
                               double budget_s, int spread, uint64_t seed, int32_t *plies_out, double *secs_out,
                               int64_t *n_evals, int *threads_used, int64_t *games_ended);

#ifdef __cplusplus
}
#endif
#endif
