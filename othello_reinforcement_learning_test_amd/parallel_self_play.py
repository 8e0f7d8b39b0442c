"""BatchMCTS / ParallelSelfPlayWorker / create_parallel_self_play_worker -- mirrors of
/root/reference/src/train/parallel_self_play.py:31-434, backed by the HIP engine.

The reference's schedule (N lock-step games, one leaf per game per simulation step, one batched
network call per step) is exactly the device schedule: one wavefront per game, one network launch
per simulation step over the dense leaf batch.

Parallel-worker semantics kept: the search always runs at temperature 1 and the
stored pi is always the visit distribution; argmax only picks the move after the threshold.
"""
import os
import time

import numpy as np

from .engine import HipResNetEvaluator, LaneOverlapCheck, SearchEngine, lane_streams
from .self_play import tuples_from_arrays


class BatchMCTS:
    """search_batch(boards, num_simulations, temperature, add_dirichlet_noise) -> [(pi, value)]
    (parallel_self_play.py:80-170)."""

    def __init__(self, model, device=None, c_puct=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                 precision=None, evaluator=None):
        self.model = model
        self.device = device
        self.c_puct = c_puct
        self.dirichlet_alpha = dirichlet_alpha
        self.dirichlet_epsilon = dirichlet_epsilon
        # evaluator: a HipResNetEvaluator (default: built from `model`), or any object with
        # host_eval(self u64[m], opp u64[m], legal u64[m]) -> (probs f32[m,65], values f32[m]) -- an external
        # evaluator driven through oth_search_leaves / oth_search_expand (stub evaluators of the parity tests)
        self.evaluator = evaluator or HipResNetEvaluator(model, precision=precision)
        self._external = hasattr(self.evaluator, "host_eval")
        self._engines = {}

    def _engine(self, n_boards, num_simulations):
        cap = 1
        while cap < n_boards:
            cap *= 2
        key = (cap, int(num_simulations))
        eng = self._engines.get(key)
        if eng is None:
            eng = SearchEngine(cap, num_simulations, c_puct=self.c_puct,
                               dirichlet_alpha=self.dirichlet_alpha,
                               dirichlet_epsilon=self.dirichlet_epsilon,
                               evaluator=None if self._external else self.evaluator)
            self._engines[key] = eng
        return eng

    def search_batch(self, boards, num_simulations, temperature=1.0, add_dirichlet_noise=False):
        n = len(boards)
        if n == 0:
            return []
        if not self._external:
            self.evaluator.refresh()
        if add_dirichlet_noise:  # :111-118: one draw per board, in board order
            for b in boards:
                np.random.dirichlet([self.dirichlet_alpha] * len(b.get_legal_moves()))
        eng = self._engine(n, num_simulations)
        sb, ob = [b.self_board for b in boards], [b.opp_board for b in boards]
        if self._external:
            pi, _, _, _ = eng.search_with(sb, ob, self.evaluator.host_eval, float(temperature))
        else:
            eng.search_begin(sb, ob)
            eng.search_run_rescued()
            pi, _, _, _ = eng.search_results(float(temperature))
        return [(pi[i].copy(), 0.0) for i in range(n)]  # root value is always 0.0


class ParallelSelfPlayWorker:
    def __init__(self, board_class, model, device=None, num_simulations=25, temperature_threshold=15,
                 num_parallel_games=8, c_puct=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                 rng_mode=None, precision=None, verbose=True, eval_cache_log2=0, lanes=None, continuous=False,
                 stagger_rounds=0, device_slots=None):
        self.board_class = board_class
        self.num_simulations = num_simulations
        self.temperature_threshold = temperature_threshold
        self.num_parallel_games = num_parallel_games
        self.rng_mode = rng_mode or os.environ.get("OTHELLO_AMD_RNG", "device")
        if self.rng_mode not in ("device", "numpy"):
            raise ValueError("rng_mode must be 'device' or 'numpy'")
        self.verbose = verbose
        self.batch_mcts = BatchMCTS(model, device, c_puct, dirichlet_alpha, dirichlet_epsilon,
                                    precision=precision)
        # device_slots: game slots of the engine in device-RNG mode.  The reference's num_parallel_games (8-32 in its
        # configs, parallel_self_play.py:232) is how many games share one network batch; with the counter-based RNG a
        # game's tuples depend on (seed, game id) only, so playing MORE games of a call at once changes nothing but the
        # speed (42 games/s at 8 slots, >500 at 128).  None = as many slots as a call has episodes, at least
        # num_parallel_games and at most 4096 (the engine grows on demand); an int pins it (num_parallel_games = the
        # reference's width).  The numpy-RNG mode always plays num_parallel_games lock-step games, as the reference does.
        self.device_slots = None if device_slots is None else int(device_slots)
        self._engine_args = dict(temperature_threshold=temperature_threshold, c_puct=c_puct,
                                 dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon,
                                 store_late_onehot=False, eval_cache_log2=eval_cache_log2)
        self.engine = SearchEngine(self.device_slots or num_parallel_games, num_simulations,
                                   temperature_threshold=temperature_threshold, c_puct=c_puct,
                                   dirichlet_alpha=dirichlet_alpha, dirichlet_epsilon=dirichlet_epsilon,
                                   store_late_onehot=False, evaluator=self.batch_mcts.evaluator,
                                   eval_cache_log2=eval_cache_log2)
        # lanes > 1 (device RNG mode): the slots are split into independent groups, each an engine on its own
        # stream and host thread; their kernels overlap on the device (one group's network launches fill the
        # other's tails and tree-search phases: +4 % at 4096 slots).  Results are concatenated lane by lane.
        if lanes is None:   # measured: two lanes pay off once each still fills the chip (2048 slots = 1024 workgroups)
            lanes = 2 if num_parallel_games >= 4096 else 1
        self.lanes = max(1, int(lanes))
        self._lane_engines = [self.engine]
        if self.lanes > 1:
            per = max(1, num_parallel_games // self.lanes)
            self._lane_engines = [SearchEngine(per, num_simulations, temperature_threshold=temperature_threshold,
                                               c_puct=c_puct, dirichlet_alpha=dirichlet_alpha,
                                               dirichlet_epsilon=dirichlet_epsilon, store_late_onehot=False,
                                               evaluator=self.batch_mcts.evaluator, eval_cache_log2=eval_cache_log2)
                                  for _ in range(self.lanes)]
        # continuous=True (device RNG only): the slots keep playing BETWEEN execute_episodes calls (oth_stream_*): a call
        # returns the games that finished during it -- at least num_episodes -- and leaves the others in flight, so
        # no call pays for a ragged tail.  Opt-in because a game can then span a weight update of the trainer, which
        # the reference's call-by-call worker never does (parallel_self_play.py:300-316 starts every batch afresh).
        self.continuous = bool(continuous)
        self.stagger_rounds = int(stagger_rounds)
        self._streaming = False
        self.last_game_ids = None
        if self.continuous and self.rng_mode != "device":
            raise ValueError("continuous=True needs rng_mode='device'")
        self.last_stats = {}
        self._ran = [self.engine]   # engines the last device-RNG call ran on
        # lanes > 1: whether the lanes' network launches really overlap is CHECKED on the first multi-lane call (HIP-event hooks
        # on for that call; engine.LaneOverlapCheck): a serialised arrangement (two streams on one hardware queue, -8...-10 %)
        # is announced once and the streams are drawn again for the next call
        self._lane_check = None

    # ---- device RNG: whole call on the GPU, finished slots refilled -------------------------
    def _grow_engine(self, num_episodes):
        """Auto mode: make the single engine wide enough to play the whole call at once (power of two, <= 4096)."""
        if self.device_slots is not None:
            return
        if self._streaming:   # in-flight games live in the current engine: never replace it under a stream
            return
        want = self.num_parallel_games
        while want < min(int(num_episodes), 4096):
            want *= 2
        want = min(want, max(4096, self.num_parallel_games))
        if want > self.engine.max_games:
            self.engine = SearchEngine(want, self.num_simulations, evaluator=self.batch_mcts.evaluator,
                                       **self._engine_args)

    def _run_device(self, num_episodes, add_dirichlet_noise):
        seed = int(np.random.randint(0, 2**62))
        if self.lanes == 1 or num_episodes < 2 * self.lanes:
            self._grow_engine(num_episodes)
            self._ran = [self.engine]
            n = self.engine.selfplay_run_rescued(num_episodes, seed, add_dirichlet_noise)
            return self.engine.selfplay_fetch(n)[:3]
        self._ran = self._lane_engines
        import threading

        import torch
        shares = [num_episodes // self.lanes + (1 if k < num_episodes % self.lanes else 0) for k in range(self.lanes)]
        out = [None] * self.lanes

        dev = torch.cuda.current_device()
        streams = lane_streams(self.lanes, dev)   # made once per process: new streams per call can share a hardware queue
        if self._lane_check is None:
            self._lane_check = LaneOverlapCheck(self.lanes, dev)
        measuring = self._lane_check.pending
        if measuring:
            self._lane_check.begin(self._lane_engines)

        errors = []

        def run(k):
            try:
                torch.cuda.set_device(dev)   # per-thread state: a new thread starts on device 0
                with torch.cuda.stream(streams[k]):
                    eng = self._lane_engines[k]
                    n = eng.selfplay_run(shares[k], seed + 7919 * (k + 1), add_dirichlet_noise)
                    out[k] = eng.selfplay_fetch(n)[:3]
            except BaseException as exc:   # re-raised in the calling thread: a failed lane fails the call
                errors.append(exc)
        while True:
            threads = [threading.Thread(target=run, args=(k,)) for k in range(self.lanes)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            # the lanes share one evaluator: the rescue of a saturated launch is decided here, where all of them have
            # joined (no launch in flight), and every lane then restarts from its seed
            if not self.batch_mcts.evaluator.needs_rescue():
                break
        if measuring:
            self._lane_check.end(self._lane_engines)   # (the next call fetches its streams anew)
        return tuple(np.concatenate([o[j] for o in out]) for j in range(3))

    def _run_stream(self, num_episodes):
        """continuous mode, single engine: -> arrays of every game that finished during this call (>= num_episodes)."""
        eng = self.engine
        self._ran = [eng]
        if not self._streaming:
            self.stream_seed = int(np.random.randint(0, 2**62))
            self._hist_games = max(8 * eng.max_games, 2 * int(num_episodes) + 4 * eng.max_games)
            eng.stream_begin(self.stream_seed, stagger_rounds=self.stagger_rounds, hist_games=self._hist_games)
            self._hist_games = 1 << (self._hist_games - 1).bit_length()   # the engine rounds the ring up to a power of two
            self._streaming = True
        # the history ring was sized by the first call: a later, larger request is played in steps the ring can hold
        cap = max(1, (self._hist_games - 2 * eng.max_games) // 2)
        left, parts, ids = int(num_episodes), [], []
        while left > 0:
            g, n = eng.stream_step_rescued(min(left, cap))
            ids.append(eng.game_ids())
            parts.append(eng.selfplay_fetch(n)[:3])
            left -= g
        self.last_game_ids = np.concatenate(ids)
        if len(parts) == 1:
            return parts[0]
        return tuple(np.concatenate([p[j] for p in parts]) for j in range(3))

    # ---- numpy RNG: the reference's lock-step batches, draws in the reference's order --------
    def _execute_batch_numpy(self, batch_size, add_dirichlet_noise):
        """parallel_self_play.py:324-407 with the search, recording and moves on the device; the
        host only draws the random numbers (it mirrors the boards to know the legal-move counts)."""
        eng = self.engine
        if eng.board_size != 8:   # the host mirror boards below are the reference's 8x8 OthelloBitboard
            raise ValueError("rng_mode='numpy' replays the reference's 8x8 RNG order; use rng_mode='device' on a "
                             "%dx%d board" % (eng.board_size, eng.board_size))
        boards = [self.board_class() for _ in range(batch_size)]
        for b in boards:
            b.reset()
        finished = [False] * batch_size
        ply = [0] * batch_size
        eng.selfplay_begin(batch_size)
        alpha = self.batch_mcts.dirichlet_alpha
        while not all(finished):
            active = [i for i in range(batch_size) if not finished[i]]
            if add_dirichlet_noise:  # drawn inside search_batch, board by board (:111-118)
                for i in active:
                    np.random.dirichlet([alpha] * len(boards[i].get_legal_moves()))
            pi, _ = eng.selfplay_search_rescued()
            actions = np.zeros(batch_size, dtype=np.int32)
            for i in active:  # :375-397
                if ply[i] < self.temperature_threshold:
                    a = int(np.random.choice(eng.npol, p=pi[i]))
                else:
                    a = int(np.argmax(pi[i]))
                actions[i] = a
                boards[i].make_move(a)
                ply[i] += 1
                if boards[i].is_terminal():
                    finished[i] = True
            eng.selfplay_apply(actions)
        n = eng.selfplay_end()
        return eng.selfplay_fetch(n)[:3]

    def execute_episodes(self, num_episodes, add_dirichlet_noise=True):
        """-> [(state, pi, z), ...] game-major (parallel_self_play.py:282-322)."""
        if num_episodes <= 0:
            return []
        self.batch_mcts.evaluator.refresh()
        t0 = time.time()
        if self.rng_mode == "device":
            if self.continuous:
                states, pis, zs = self._run_stream(num_episodes)
            else:
                states, pis, zs = self._run_device(num_episodes, add_dirichlet_noise)
            data = tuples_from_arrays(states, pis, zs)
        else:
            data = []
            done = 0
            while done < num_episodes:  # :300-316
                bs = min(self.num_parallel_games, num_episodes - done)
                states, pis, zs = self._execute_batch_numpy(bs, add_dirichlet_noise)
                data.extend(tuples_from_arrays(states, pis, zs))
                done += bs
        self.batch_mcts.evaluator.check_saturation()   # last line of defence: every path above rescues a saturated launch
        dt = time.time() - t0
        counters = {}
        for eng in (self._ran if self.rng_mode == "device" else [self.engine]):
            for k, v in eng.counters().items():
                counters[k] = counters.get(k, 0) + v
        games = len(self.last_game_ids) if (self.continuous and self.rng_mode == "device") else num_episodes
        ev = self.batch_mcts.evaluator
        # what the run's arithmetic was: the activation scale of the fp16-split trunk and every rescue so far
        counters.update({"precision": getattr(ev, "precision", None), "act_scale": getattr(ev, "act_scale", None),
                         "rescues": len(getattr(ev, "rescues", ()))})
        if self._lane_check is not None and self._ran is self._lane_engines:
            counters.update(self._lane_check.report())
        # the engine's counters are cumulative ("games" = every game finished so far): kept under "engine_*" names
        self.last_stats = {**{"engine_" + k if k == "games" else k: v for k, v in counters.items()},
                           "games": games, "samples": len(data), "seconds": dt,
                           "games_per_s": games / dt if dt > 0 else float("inf")}
        if self.verbose:
            print("  Self-Play: %d/%d games | %s samples | %.1fs (%.2f games/s)" %
                  (games, num_episodes, format(len(data), ","), dt, self.last_stats["games_per_s"]))
        return data

    def execute_episodes_arrays(self, num_episodes, add_dirichlet_noise=True, seed=None):
        """Array form for callers that keep replay data on the device side: (states, pis, zs)."""
        if self._streaming:
            raise RuntimeError("this worker is streaming (continuous=True): a batch run would drop its in-flight "
                               "games -- use execute_episodes")
        self.batch_mcts.evaluator.refresh()
        if seed is None:
            seed = int(np.random.randint(0, 2**62))
        self._grow_engine(num_episodes)
        n = self.engine.selfplay_run_rescued(num_episodes, seed, add_dirichlet_noise)
        return self.engine.selfplay_fetch(n)[:3]


def create_parallel_self_play_worker(config, model, device=None, **kwargs):
    """Build from a reference YAML config dict (parallel_self_play.py:410-434); same keys, same
    defaults."""
    from .bitboard import OthelloBitboard
    mcts = config.get("mcts", {})
    sp = config.get("self_play", {})
    # `self_play.continuous: true` -- a key of THIS package, absent from the reference's YAML files (absent = off = the
    # reference's call-by-call behaviour): the worker's slots keep playing between execute_episodes calls, so a trainer that
    # asks for 50-200 episodes per iteration (training.self_play_episodes_per_iter: 100 in configs/default_8x8.yaml, 50 in fast_8x8) is
    # served from full slots instead of a ragged small batch
    # (INTEGRATION.md section 1 has the rates); the one semantic difference is that a game may span a weight update.
    # `self_play.stagger_rounds` spreads the slots' start over that many ply rounds.  A keyword argument of the same name wins.
    if "continuous" in sp:
        kwargs.setdefault("continuous", bool(sp["continuous"]))
    if "stagger_rounds" in sp:
        kwargs.setdefault("stagger_rounds", int(sp["stagger_rounds"]))
    # `self_play.device_slots`: game slots on the device (default: as many as a call has episodes, see __init__).  In continuous
    # mode it is the number of games in flight: a game then spans about device_slots / num_episodes weight updates.
    if "device_slots" in sp:
        kwargs.setdefault("device_slots", int(sp["device_slots"]))
    return ParallelSelfPlayWorker(
        board_class=OthelloBitboard,
        model=model,
        device=device,
        num_simulations=mcts.get("num_simulations", 25),
        temperature_threshold=sp.get("temperature_threshold", 15),
        num_parallel_games=sp.get("num_parallel_games", 8),
        c_puct=mcts.get("c_puct", 1.0),
        dirichlet_alpha=mcts.get("dirichlet_alpha", 0.3),
        dirichlet_epsilon=mcts.get("dirichlet_epsilon", 0.25),
        **kwargs,
    )
