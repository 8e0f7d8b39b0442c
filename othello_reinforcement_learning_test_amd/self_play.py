"""SelfPlayWorker -- mirror of /root/reference/src/train/self_play.py:25-163.

``execute_episodes(num_episodes, add_dirichlet_noise)`` returns the reference's list of
``(state float32 (3,8,8), pi float32 (65,), z float)`` tuples, episode after episode.

Two random-number modes:
  * ``rng_mode="device"`` (default): all episodes of a call are played concurrently on the GPU
    (they are independent), actions sampled by the engine's counter-based RNG.  The per-call seed is
    drawn from numpy's global RNG, so ``np.random.seed`` (reference main.py:69-70) still makes a
    run reproducible.
  * ``rng_mode="numpy"``: the reference's exact loop, one game at a time, ``np.random.dirichlet`` /
    ``np.random.choice`` drawn in the reference's order: a seeded run reproduces the reference's
    (state, pi, z) stream (up to floating-point differences of the evaluator).
Serial-worker semantics kept: the stored pi is the temperature-adjusted one (one-hot
after ``temperature_threshold`` plies); z = winner_at_end * player with the reference's sign (L16).
"""
import os
from dataclasses import dataclass

import numpy as np

from .engine import SearchEngine

MAX_CONCURRENT_GAMES = 4096


@dataclass
class GameStep:  # self_play.py:17-22
    state: np.ndarray
    policy: np.ndarray
    player: int


def tuples_from_arrays(states, pis, zs, block=512):
    """Arrays -> the list of (state, pi, z) tuples the trainer/ReplayBuffer keep (buffer.py:45).  The reference hands out
    fresh arrays per tuple; 2 x 250 000 small copies per 4096 games would dominate the call, and plain row views of the
    fetch would let one surviving tuple pin the whole 250 MB of it in a long-lived deque.  Middle way: the rows are
    copied in blocks of `block` (512 x 1 KB; +0.07 s per 250 000 tuples over plain views) and each tuple holds row views of
    its block, so a surviving tuple pins 0.5 MB; z is a Python float as in the reference (self_play.py:127)."""
    out = []
    zl = zs.tolist()
    for i in range(0, len(zl), block):
        sb, pb = states[i:i + block].copy(), pis[i:i + block].copy()
        out.extend(zip(sb, pb, zl[i:i + block]))
    return out


class SelfPlayWorker:
    def __init__(self, board_class, mcts, num_simulations=25, temperature_threshold=15, rng_mode=None):
        self.board_class = board_class
        self.mcts = mcts
        self.num_simulations = num_simulations
        self.temperature_threshold = temperature_threshold
        self.rng_mode = rng_mode or os.environ.get("OTHELLO_AMD_RNG", "device")
        if self.rng_mode not in ("device", "numpy"):
            raise ValueError("rng_mode must be 'device' or 'numpy'")
        self._engine = None

    # ---- reference-order loop (numpy RNG) --------------------------------------------------
    def execute_episode(self, add_dirichlet_noise=True):
        """One game, the reference's loop (self_play.py:52-135)."""
        board = self.board_class()
        board.reset()
        history = []
        ply = 0
        while not board.is_terminal():
            player = 1 if ply % 2 == 0 else -1
            temperature = 1.0 if ply < self.temperature_threshold else 0.0
            state = board.get_tensor_input()
            policy, _ = self.mcts.search(board, self.num_simulations, temperature, add_dirichlet_noise)
            history.append(GameStep(state.copy(), policy.copy(), player))
            if temperature == 0:
                action = int(np.argmax(policy))
            else:
                action = np.random.choice(len(policy), p=policy)
            board.make_move(action)
            ply += 1
        winner = board.get_winner()
        return [(s.state, s.policy, float(winner * s.player)) for s in history]

    # ---- all episodes at once on the device ------------------------------------------------
    def _device_engine(self, num_episodes):
        g = min(int(num_episodes), MAX_CONCURRENT_GAMES)
        if self._engine is None or self._engine.max_games < g or \
                self._engine.num_simulations != self.num_simulations:
            self._engine = SearchEngine(
                g, self.num_simulations, temperature_threshold=self.temperature_threshold,
                c_puct=self.mcts.c_puct, dirichlet_alpha=self.mcts.dirichlet_alpha,
                dirichlet_epsilon=self.mcts.dirichlet_epsilon, store_late_onehot=True,
                evaluator=self.mcts.evaluator)
        return self._engine

    def execute_episodes(self, num_episodes, add_dirichlet_noise=True):
        if num_episodes <= 0:
            return []
        if self.rng_mode == "numpy":
            data = []
            for _ in range(num_episodes):  # self_play.py:154-161
                data.extend(self.execute_episode(add_dirichlet_noise))
            return data
        self.mcts.evaluator.refresh()
        eng = self._device_engine(num_episodes)
        seed = int(np.random.randint(0, 2**62))
        n = eng.selfplay_run_rescued(num_episodes, seed, add_dirichlet_noise)
        states, pis, zs, _ = eng.selfplay_fetch(n)
        return tuples_from_arrays(states, pis, zs)


def augment_data_with_symmetries(training_data, board_class):
    """The reference function of this name returns its input unchanged (self_play.py:166-212 appends
    only the originals); kept identical so a caller sees the same data volume.  The real 8-fold
    transform is ``OthelloBitboard.get_symmetries``."""
    return [(s, p, v) for (s, p, v) in training_data]
