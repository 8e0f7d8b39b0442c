/*
 * othello_mi355x.h -- C ABI of libothello_mi355x.so: the MI355X (gfx950) self-play hot path of an
 * AlphaZero Othello trainer.  Plain pointers and sizes only; no torch types.
 *
 * The reference (Sylphy0052/Othello_Reinforcement_learning_test) has no FFI for this path: its
 * boundary is a set of duck-typed Python objects.  Each entry point below names
 * the reference interface it replaces (paths relative to the reference root).  The Python mirror of
 * those objects lives in the othello_reinforcement_learning_test_amd package and binds this ABI with ctypes;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returning int returns OTH_OK (0) or a negative OTH_E_* code;
 *     oth_last_error() gives the message (thread-local).  Nothing throws, nothing aborts.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All device work is
 *     enqueued on it; functions documented as "synchronous" wait for it before returning.
 *   - pointers marked DEVICE must be device-accessible (hipMalloc / torch.cuda tensors); pointers
 *     marked HOST are ordinary host memory; ANY accepts both (hipMemcpyDefault).
 *   - the caller owns every buffer it passes; handles are opaque and freed by their destroy call.
 *   - there is no CPU fallback: without a gfx950 device every device entry point fails with
 *     OTH_E_NO_DEVICE.  The single-board host functions (section 1) are host code by nature.
 */
#ifndef OTHELLO_MI355X_H
#define OTHELLO_MI355X_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OTH_OK 0
#define OTH_E_NO_DEVICE (-1)
#define OTH_E_INVALID (-2)
#define OTH_E_HIP (-3)
#define OTH_E_STATE (-4)
#define OTH_E_UNSUPPORTED (-5)

const char *oth_last_error(void);
/* 1 if a gfx950 device is usable, else 0 (never fails) */
int oth_device_available(void);
const char *oth_version(void);

/* =============================================================================================
 * 1. Single-board rules (host side of the OthelloBitboard object, src/cython/bitboard.pxd:11-48)
 * =========================================================================================== */
typedef struct {
    uint64_t self_board; /* bitboard.pxd:25 */
    uint64_t opp_board;  /* bitboard.pxd:26 */
    int32_t move_count;  /* bitboard.pxd:27 */
    int32_t passed;      /* bitboard.pxd:28 */
} oth_board;

void oth_board_reset(oth_board *b);                                    /* bitboard.pyx:52   reset */
uint64_t oth_legal_moves(uint64_t self_board, uint64_t opp_board);     /* pyx:135/:187 get_legal_moves_bits */
uint64_t oth_flip_bits(int pos, uint64_t self_board, uint64_t opp_board); /* pyx:116 _get_flip_bits */
int oth_board_make_move(oth_board *b, int pos);  /* pyx:195 make_move: 1 ok, 0 invalid (state unchanged) */
int oth_board_is_terminal(const oth_board *b);                         /* pyx:249 is_terminal */
int oth_board_get_winner(const oth_board *b);                          /* pyx:266 get_winner */
void oth_board_get_tensor_input(const oth_board *b, float *out192 /*HOST*/); /* pyx:300 get_tensor_input */
/* pyx:338 get_symmetries: 8 x ((3,8,8) state, (65,) pi) */
void oth_board_get_symmetries(const oth_board *b, const float *pi65, float *states8x192, float *pis8x65);

/* =============================================================================================
 * 2. Batched rules on the device (the same functions over arrays of positions; K2 of SURVEY 2)
 * =========================================================================================== */
/* legal[i] = get_legal_moves_bits of (self[i], opp[i])                           all DEVICE */
int oth_legal_moves_batch(const uint64_t *self_b, const uint64_t *opp_b, uint64_t *legal, int64_t n, void *stream);
/* make_move per position, in place; ok[i] in {0,1}; flips[i] may be NULL.       all DEVICE */
int oth_make_move_batch(uint64_t *self_b, uint64_t *opp_b, const int32_t *pos, int32_t *ok, uint64_t *flips,
                        int64_t n, void *stream);
/* terminal[i] = is_terminal, winner[i] = get_winner                              all DEVICE */
int oth_status_batch(const uint64_t *self_b, const uint64_t *opp_b, int32_t *terminal, int32_t *winner,
                     int64_t n, void *stream);
/* out[i] = get_tensor_input as float32 [n,3,8,8]                                 all DEVICE */
int oth_tensor_input_batch(const uint64_t *self_b, const uint64_t *opp_b, float *out, int64_t n, void *stream);
/* rules checksum over the LCG position stream of tests/golden (size-independent parity property) */
int oth_rules_checksum(int64_t n, uint64_t *legal_acc /*HOST*/, uint64_t *flip_acc /*HOST*/, void *stream);

/* ---- board size 6 (BASELINE configs[4]; reference configs/debug_6x6.yaml) -------------------------------------
 * The `_n` forms take board_size = 8 or 6.  8 is the reference's game (identical to the functions above).  6 is the
 * SAME algorithm on a 6x6 grid: bit i = row*6 + col, pass action 36, 37-entry policies, planes [3,6,6]; the eight
 * ray directions and the reference's post-shift edge masks are carried over (bitboard.pyx:20-38 with N = 6).  The
 * reference itself has no 6x6 rules (its game.size is never read), so 6x6 results are PARITY UNPINNED: they are
 * checked against the 6x6 build of the CPU oracle, not against the reference. */
void oth_board_reset_n(int board_size, oth_board *b);
uint64_t oth_legal_moves_n(int board_size, uint64_t self_board, uint64_t opp_board);
uint64_t oth_flip_bits_n(int board_size, int pos, uint64_t self_board, uint64_t opp_board);
int oth_board_make_move_n(int board_size, oth_board *b, int pos);
int oth_board_is_terminal_n(int board_size, const oth_board *b);
int oth_legal_moves_batch_n(int board_size, const uint64_t *self_b, const uint64_t *opp_b, uint64_t *legal, int64_t n,
                            void *stream);
int oth_make_move_batch_n(int board_size, uint64_t *self_b, uint64_t *opp_b, const int32_t *pos, int32_t *ok,
                          uint64_t *flips, int64_t n, void *stream);
int oth_status_batch_n(int board_size, const uint64_t *self_b, const uint64_t *opp_b, int32_t *terminal, int32_t *winner,
                       int64_t n, void *stream);
int oth_tensor_input_batch_n(int board_size, const uint64_t *self_b, const uint64_t *opp_b, float *out, int64_t n,
                             void *stream);
int oth_rules_checksum_n(int board_size, int64_t n, uint64_t *legal_acc /*HOST*/, uint64_t *flip_acc /*HOST*/, void *stream);

/* =============================================================================================
 * 3. Evaluator: OthelloResNet forward (src/model/net.py:139-205), eval mode
 * =========================================================================================== */
typedef struct oth_net oth_net;
#define OTH_PREC_F32 0     /* exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32): 16/32/64/128 filters, 8x8 and 6x6 */
#define OTH_PREC_F16X3 1   /* MFMA, fp16 hi/lo split of both operands, fp32 accumulate (fp32-equivalent): 32 / 64 / 128
                              filters on 8x8 and 6x6 */
#define OTH_PREC_F16 2     /* MFMA, single fp16 pass (fast; NOT within the 1e-4 parity tolerance in general); 128 filters, 8x8 */
#define OTH_PREC_F16X3_DIRECT 3 /* the same fp16 hi/lo split arithmetic as OTH_PREC_F16X3 on the DIRECT 3x3 convolution kernel
                                 * (k_trunk16) instead of the 1-D Winograd one (k_trunk_w): 128 filters on 8x8 only; the A/B
                                 * partner of the default and an independent cross-check of it (same tolerance, different
                                 * summation) */

/* net.py:157-180 __init__(num_blocks, num_filters, board_size).  board_size 8 or 6 (configs/debug_6x6.yaml);
 * num_filters 16, 32, 64 or 128.  A 6x6 network takes positions as bit i = row*6 + col (i < 36) and returns
 * 37 log-probabilities per position. */
oth_net *oth_net_create(int num_blocks, int num_filters, int board_size);
/* board_size^2 + 1: length of a policy row of this network (65 or 37) */
int oth_net_policy_size(const oth_net *net);
void oth_net_destroy(oth_net *net);
/* number of float32 values oth_net_load_state expects */
int64_t oth_net_state_floats(const oth_net *net);
/* Load weights = model.state_dict() flattened: every floating tensor in registration order
 * (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, ... fc.weight, fc.bias),
 * int64 num_batches_tracked skipped.  HOST float32.  BatchNorms are folded (eval mode, L19) and
 * the weights repacked for the kernels.  Call again whenever the trainer has updated the model. */
int oth_net_load_state(oth_net *net, const float *blob, int64_t n_floats, int precision);
/* forward(x) for x given as packed bitboards (self, opp, legal planes of get_tensor_input).
 * logp: [n,65] ([n,37] for a 6x6 network) log-probabilities, v: [n].  all DEVICE.  n_valid (DEVICE int32*, may be NULL): when
 * given, only the first *n_valid (<= n) positions are evaluated (device-side batch length). */
int oth_net_forward_bits(oth_net *net, const uint64_t *self_b, const uint64_t *opp_b, const uint64_t *legal,
                         int64_t n, const int32_t *n_valid, float *logp, float *v, void *stream);
/* forward(x) for x float32 [n,3,S,S] holding 0/1 planes (the reference's input format) DEVICE */
int oth_net_forward_planes(oth_net *net, const float *x, int64_t n, float *logp, float *v, void *stream);
/* The fp16-split trunk kernels (OTH_PREC_F16X3 / OTH_PREC_F16) carry activations pre-scaled by a power of two (the
 * "activation scale", 16 by default: it keeps the low parts of the operand split clear of the f16 subnormals) and clamp the
 * scaled value at the f16 range: the direct kernels (k_trunk16, k_trunk_h3) at 60000 / scale = 3750, the Winograd trunks
 * (k_trunk_w, k_trunk_w6: the DEFAULT for 128 filters on 8x8 and 64 filters on 6x6) at 30000 / scale = 1875 (their
 * transformed operand is up to twice an activation); the reference's fp32 forward (net.py:182-205) has no clamp.
 * *flag = 1 when any launch since the last call clamped a value -- the results of those launches differ from the
 * reference's.  The rescue: halve the scale (oth_net_set_act_scale: 16 -> 8 -> ... -> 1 widens the range to 30 000 /
 * 60 000) and run the affected call again from its start state (oth_engine_snapshot / oth_engine_restore for a stream
 * step or a lock-step search; a batch run restarts from its seed); only a network that still clamps at scale 1 needs
 * OTH_PREC_F32.  The Python workers do exactly that.  Reads and clears a device flag; synchronises `stream`.  HOST flag. */
int oth_net_saturated(oth_net *net, int32_t *flag, void *stream);
/* The activation scale of the fp16-split trunks (no reference counterpart: net.py:182-205 is fp32 throughout): 1, 2, 4, 8
 * or 16 (default).  A run-time value -- it enters a launch through the stem's input planes, the biases and the heads'
 * un-scaling -- that survives oth_net_load_state.  set waits for the device to idle (hipDeviceSynchronize) and rewrites the
 * scaled biases: call it between calls, never while a launch on this network is in flight.  No effect on OTH_PREC_F32. */
int oth_net_set_act_scale(oth_net *net, float scale);
int oth_net_get_act_scale(const oth_net *net, float *scale /*HOST*/);
/* Which trunk kernel a launch of n positions of this network runs (no reference counterpart: the reference's forward,
 * net.py:182-205, is torch ops; measurement needs the kernel's name and its arithmetic factor from the library, not
 * re-derived by the caller): `name` (HOST, name_cap bytes, NUL-terminated), *issued_per_flop = MFMA FLOPs the kernel
 * issues per algorithmic FLOP of the direct 3x3 convolutions (1.0 exact fp32; 2.75 k_trunk16; 2.0 the Winograd trunks;
 * 3.0 x tile padding k_trunk_h3), *clamp = the activation clamp at the CURRENT activation scale (0 = none).  Any output
 * pointer may be NULL.  Needs loaded weights; no device work. */
int oth_net_kernel_info(const oth_net *net, int64_t n, char *name, int32_t name_cap, double *issued_per_flop,
                        double *clamp);

/* probs[i] = exp(logp[i]) exactly as the engine's expansion computes it from the network's log-probabilities
 * (mcts.py:189 / parallel_self_play.py:72-76 `policy_probs = torch.exp(policy_logits)`): lets an external
 * evaluator driven through oth_search_expand(is_log = 0) see the same float32 priors.  all DEVICE. */
int oth_policy_exp(const float *logp, float *probs, int64_t n, void *stream);

/* =============================================================================================
 * 4. Search + self-play engine
 *    replaces MCTS (src/mcts/mcts.py:17), MCTSNode (src/mcts/node.py:12), BatchMCTS
 *    (src/train/parallel_self_play.py:31) and the worker loops (self_play.py:52-163,
 *    parallel_self_play.py:282-407).
 * =========================================================================================== */
typedef struct oth_engine oth_engine;

typedef struct {
    int32_t max_games;         /* G: concurrent game slots (num_parallel_games, parallel_self_play.py:232) */
    int32_t num_simulations;   /* mcts.num_simulations (0..4000) */
    int32_t temperature_threshold; /* self_play.temperature_threshold */
    float c_puct;              /* python float -> float32 weak scalar at node.py:116 */
    double dirichlet_alpha;    /* mcts.py:33 */
    double dirichlet_epsilon;  /* mcts.py:34 */
    int32_t store_late_onehot; /* 1: SelfPlayWorker semantics (pi stored one-hot after the threshold, L14);
                                  0: ParallelSelfPlayWorker semantics (always the visit distribution, L15) */
    int32_t eval_cache_log2;   /* 0 = off (default).  >0: transposition table of 2^n network results (2^(n-1) sets of two
                                  ways, 280 B per entry) keyed by position, reused bit-identically within a run (cleared at
                                  every run start / stream step);
                                  skips re-evaluating positions the search has already evaluated (no reference
                                  counterpart: the reference re-evaluates; outputs are identical) */
    int32_t board_size;        /* 0 or 8: the reference's 8x8 game.  6: the same engine on a 6x6 board (BASELINE
                                  configs[4]): pass action 36; every [.,65] array below is [.,37] and every
                                  [.,3,8,8] array [.,3,6,6]; rules as oth_*_n(6, ...) define them -- PARITY
                                  UNPINNED, the reference has no 6x6 rules.  The network given to
                                  oth_engine_set_net must have been created for the same board size. */
} oth_engine_cfg;

oth_engine *oth_engine_create(const oth_engine_cfg *cfg);
void oth_engine_destroy(oth_engine *e);
/* evaluator used by oth_search_run / oth_selfplay_run (not owned) */
int oth_engine_set_net(oth_engine *e, oth_net *net);

/* ---- step-wise search over n <= max_games root positions (BatchMCTS.search_batch semantics:
 *      one leaf per game per simulation step).  An external evaluator can be driven through
 *      leaves()/expand(); the built-in network through oth_search_run. -------------------------- */
/* roots: HOST arrays self[n], opp[n].  Resets the trees and queues the n roots for evaluation. */
int oth_search_begin(oth_engine *e, const uint64_t *self_b, const uint64_t *opp_b, int32_t n, void *stream);
/* select one leaf per game (node.py:91 select_child, parallel_self_play.py:172 _select_leaf);
 * terminal leaves are backed up at once with float(get_winner()) (mcts.py:127-130). */
int oth_search_select(oth_engine *e, void *stream);
/* positions waiting for evaluation after begin/select.  Synchronous.  count: HOST int32;
 * self/opp/legal: HOST arrays of capacity max_games (may be NULL). */
int oth_search_leaves(oth_engine *e, int32_t *count, uint64_t *self_b, uint64_t *opp_b, uint64_t *legal, void *stream);
/* Feed evaluator results for the pending positions, in the order oth_search_leaves gave them, expand
 * (node.py:62) and back up (mcts.py:152).  policy: ANY float32 [count,65]; is_log=1: log-probs (the
 * network's output; exp applied as mcts.py:189), 0: probabilities.  value: ANY float32 [count]. */
int oth_search_expand(oth_engine *e, const float *policy, const float *value, int32_t is_log, void *stream);
/* begin(already called) + num_simulations x (select, network, expand) with the engine's network */
int oth_search_run(oth_engine *e, void *stream);
/* Results per root i.  Synchronous.  Any output may be NULL.  HOST.
 *   pi[n,65]      node.py:147 get_policy_distribution(temperature); temperature 0 or 1 (any other value is an
 *                 error here: the Python mirror evaluates node.py:175-177's counts ** (1/T) on the host from visits[])
 *   visits[n,65]  child visit counts;  value_sum[n,65] child W (float64);  prior[n,65] child P */
int oth_search_results(oth_engine *e, double temperature, float *pi, int32_t *visits, double *value_sum,
                       float *prior, void *stream);

/* ---- self-play -------------------------------------------------------------------------------
 * Plays num_games complete games on the device, max_games at a time with finished slots refilled,
 * sampling actions with a counter-based RNG keyed by (seed, game id, ply).  add_noise is accepted
 * for interface parity; Dirichlet noise on root priors cannot change any output of this search
 * (the root is never backed up, so its exploration term is zero: SURVEY L10/L12).
 * Synchronous.  Results stay in the engine until the next run; read them with oth_selfplay_fetch. */
int oth_selfplay_run(oth_engine *e, int32_t num_games, uint64_t seed, int32_t add_noise, int64_t *n_samples /*HOST*/,
                     void *stream);
/* Lock-step variant driven ply by ply from the host (reference RNG compatibility: the caller draws
 * numpy's dirichlet/choice itself).  begin: start n <= max_games games from the initial position. */
int oth_selfplay_begin(oth_engine *e, int32_t n, void *stream);
/* search the current positions of all unfinished games (roots + sims with the engine's network);
 * pi: HOST [n,65] visit distributions (T=1), active: HOST [n] 1 for games that were searched. */
int oth_selfplay_search(oth_engine *e, float *pi, int32_t *active, void *stream);
/* record (state, pi, player) and play actions[i] (HOST [n], ignored for finished games) */
int oth_selfplay_apply(oth_engine *e, const int32_t *actions, int32_t *n_unfinished /*HOST*/, void *stream);
/* finish a lock-step run: assign z and compact.  n_samples: HOST */
int oth_selfplay_end(oth_engine *e, int64_t *n_samples, void *stream);
/* ---- streaming self-play (steady state: the slots stay full ACROSS calls) ----------------------
 * The reference's trainer calls execute_episodes(num_episodes) once per iteration (src/train/trainer.py:180-185)
 * and each call starts from empty slots.  A stream keeps the max_games slots playing between calls: a step
 * returns the games that FINISHED during it and leaves the others in flight, so there is no ragged tail per
 * call.  Every game is still one complete self-play episode from the initial position with the same per-game
 * arithmetic as oth_selfplay_run (random draws keyed by (seed, game id, ply), so a game's tuples depend on its id
 * only, not on the slot or the step it lands in).
 *   begin: all slots idle; slot g starts game id g at ply round g * stagger_rounds / max_games (0: all at once;
 *          ~61 spreads the game phases evenly so that games finish at a steady rate), later games get ids
 *          max_games, max_games+1, ... in the order slots free up.  hist_games = capacity of the history ring
 *          (0: 8 * max_games); a step can return at most hist_games - 2 * max_games games.
 *   step:  run ply rounds until at least min_games games have finished since the last step (checked one round
 *          late so the device never idles: the step ends with the round after the one that reached the target),
 *          then compact the tuples of ALL games finished so far, ascending game id.  Synchronous.  Read them with
 *          oth_selfplay_fetch / oth_selfplay_device_ptrs; oth_selfplay_game_ids gives their ids.
 * oth_engine_counters are cumulative over the stream.  The network's weights may be reloaded between steps (games
 * in flight then continue with the new weights; the optional evaluation cache is cleared at every step). */
int oth_stream_begin(oth_engine *e, uint64_t seed, int32_t stagger_rounds, int32_t hist_games, void *stream);
int oth_stream_step(oth_engine *e, int32_t min_games, int32_t *n_games /*HOST*/, int64_t *n_samples /*HOST*/,
                    void *stream);
/* ids of the games whose tuples the last run / step produced, in output order.  ids: HOST [capacity] (may be NULL) */
int oth_selfplay_game_ids(oth_engine *e, int32_t *ids, int32_t capacity, int32_t *count);
/* Copy the replay tuples of the last run, game-major then ply order (the order of
 * parallel_self_play.py:400-405): states [n,3,8,8] f32, pis [n,65] f32, zs [n] f32,
 * game_len [num_games] int32 (may be NULL).  Destination pointers: ANY.  Synchronous. */
int oth_selfplay_fetch(oth_engine *e, float *states, float *pis, float *zs, int32_t *game_len, void *stream);
/* DEVICE pointers to the same compacted arrays (valid until the next run), for on-device consumers
 * such as the RCCL all-gather: no copy. */
int oth_selfplay_device_ptrs(oth_engine *e, float **states, float **pis, float **zs, int64_t *n_samples);

/* Snapshot / restore of a stream or a lock-step run BETWEEN two calls (no reference counterpart: the reference's fp32
 * network cannot saturate, parallel_self_play.py:53-78): snapshot copies the slots' game positions, the queued roots, the
 * status words, the counters and the bookkeeping of the history ring (a few MB, on `stream`) so that a step / search whose
 * network launches saturated (oth_net_saturated) can be played again from exactly the state it started in after
 * oth_net_set_act_scale: restore puts that state back, forgets the games the abandoned call finished (their ring
 * entries are free again) and clears the evaluation cache.  Games are keyed by (seed, game id, ply), so the repeated call
 * replays the same games.  (With the evaluation cache on, roots that were taken from the cache BEFORE the snapshot keep their
 * rows: outputs of earlier launches that did not saturate -- valid, though computed at the previous activation scale.)
 * snapshot: call when no step / search is in progress; restore: needs a snapshot. */
int oth_engine_snapshot(oth_engine *e, void *stream);
int oth_engine_restore(oth_engine *e, void *stream);

/* counters of the last run: [0] network evaluations, [1] simulations, [2] plies, [3] games,
 * [4] network batches launched, [5] terminal-leaf simulations, [6] evaluation-cache hits */
int oth_engine_counters(oth_engine *e, int64_t out[8]);
/* Statistics of the optional evaluation cache since the run / stream began (no reference counterpart: mcts.py:71 and
 * parallel_self_play.py:106 build a fresh tree per search and evaluate every position again -- the redundancy the cache
 * removes): out[0] distinct positions evaluated (first evaluation since the cache was last cleared: compulsory misses),
 * out[1] repeated evaluations (the position had been evaluated since the clear: its entry was replaced in between, or it
 * missed twice in one launch), out[2] inserts that replaced a live entry of ANOTHER position (conflict evictions),
 * out[3] entries of the table (0: no cache).  out[0] + out[1] = network evaluations.  Synchronises `stream`. */
int oth_engine_cache_stats(oth_engine *e, int64_t out[4], void *stream);
/* timing hook for bench.py: HIP-event time (ms) spent in the network kernel during the last run, and
 * the number of launches; measured on the stream the kernels ran on */
int oth_engine_kernel_time(oth_engine *e, double *net_ms, int64_t *net_launches, double *tree_ms, int64_t *tree_launches);
int oth_engine_set_timing(oth_engine *e, int32_t enable);
/* (start, end) in ms of every network launch of the last run on ONE process-wide HIP-event time axis, so that the
 * union of launches that overlap across engines / streams can be formed.  spans: HOST [capacity][2] (may be NULL
 * to query the count). */
int oth_engine_net_spans(oth_engine *e, double *spans, int64_t capacity, int64_t *count);

/* =============================================================================================
 * 5. Replay-tuple operations on the device
 * =========================================================================================== */
/* 8-fold dihedral augmentation = OthelloBitboard.get_symmetries (bitboard.pyx:338-370) applied to every
 * sample: out[8*i + k] is variant k of sample i (k = 2j: rot90^j; k = 2j+1: that, then left-right flip);
 * pi[64] and z are copied.  states [n,3,8,8] -> [8n,3,8,8], pis [n,65] -> [8n,65], zs [n] -> [8n].
 * all DEVICE.  (The reference's augment_data_with_symmetries, self_play.py:166-212, returns its input
 * unchanged; this is the transform it describes.) */
int oth_augment_symmetries(const float *states, const float *pis, const float *zs, int64_t n, float *states_out,
                           float *pis_out, float *zs_out, void *stream);

/* ReplayBuffer.sample's minibatch assembly (src/train/buffer.py:59-100: random.sample + np.array over the tuples)
 * as a device gather: out row i = ring row (ring_start + idx[i]) % ring_size (ring_size = 0: row idx[i]).
 * states [cap,3,8,8], pis [cap,65], zs [cap] -> states_out [n,3,8,8], pis_out [n,65], values_out [n] (the caller
 * views it as [n,1], buffer.py:83).  idx: int64 [n].  all DEVICE. */
int oth_replay_gather(const float *states, const float *pis, const float *zs, const int64_t *idx, int64_t n,
                      int64_t ring_start, int64_t ring_size, float *states_out, float *pis_out, float *values_out,
                      void *stream);

/* The same two operations for a board of board_size x board_size squares (8 or 6): rows of [3,S,S] / [S*S+1] floats
 * (OthelloResNet is size-parametric, /root/reference/src/model/net.py:81,116; BASELINE configs[4] plays 6x6). */
int oth_augment_symmetries_n(int board_size, const float *states, const float *pis, const float *zs, int64_t n,
                             float *states_out, float *pis_out, float *zs_out, void *stream);
int oth_replay_gather_n(int board_size, const float *states, const float *pis, const float *zs, const int64_t *idx,
                        int64_t n, int64_t ring_start, int64_t ring_size, float *states_out, float *pis_out,
                        float *values_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
