"""The reference's own small configurations through the drop-in worker: fast_8x8.yaml (8 parallel games, 15 sims, 10x128)
and strong_8x8.yaml (16 parallel games, 100 sims), execute_episodes(100) as trainer.py:180 calls it -- with the engine
pinned to the reference's batch width (device_slots = num_parallel_games) and with the default (grown to the call)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import othello_reinforcement_learning_test_amd as pkg

torch.manual_seed(42)
net = pkg.OthelloResNet(10, 128).eval()
for name, par, sims, episodes in (("fast_8x8", 8, 15, 100), ("strong_8x8", 16, 100, 100)):
    for slots in (par, None):
        w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=sims, temperature_threshold=15,
                                       num_parallel_games=par, verbose=False, device_slots=slots)
        np.random.seed(1)
        w.execute_episodes(par)                      # warm-up
        np.random.seed(2)
        t0 = time.time()
        data = w.execute_episodes(episodes)
        dt = time.time() - t0
        print("%-11s %3d sims, execute_episodes(%d), num_parallel_games %2d, engine slots %4d: %6.2f s = %7.1f games/s (%d tuples)"
              % (name, sims, episodes, par, w.engine.max_games, dt, episodes / dt, len(data)), flush=True)

# Round 6: what a 100-episode trainer gets from `self_play.continuous: true` (create_parallel_self_play_worker): the slots keep
# playing between calls, so a call is served from FULL slots.  Ten successive execute_episodes(100) calls per slot count (after
# one call that fills the pipeline); a game then spans about device_slots / 100 weight updates of the trainer.
print("continuous mode (self_play.continuous: true), fast_8x8 (15 sims), ten execute_episodes(100) calls per row:")
for slots in (128, 512, 2048, 4096):
    config = {"mcts": {"num_simulations": 15, "c_puct": 1.0, "dirichlet_alpha": 0.3, "dirichlet_epsilon": 0.25},
              "self_play": {"temperature_threshold": 15, "num_parallel_games": 8, "continuous": True, "stagger_rounds": 61,
                            "device_slots": slots}}
    w = pkg.create_parallel_self_play_worker(config, net, verbose=False)
    np.random.seed(3)
    w.execute_episodes(100)                          # fills the pipeline (staggered start)
    torch.cuda.synchronize()
    t0, games, tuples = time.time(), 0, 0
    for _ in range(10):
        data = w.execute_episodes(100)
        games += len(w.last_game_ids)
        tuples += len(data)
    dt = time.time() - t0
    print("  device_slots %4d: %6.2f s for 10 calls = %7.1f games/s (%d games, %d tuples; %.0f ms per call; a game spans ~%.0f calls)"
          % (slots, dt, games / dt, games, tuples, dt * 100, slots / max(1.0, games / 10.0)), flush=True)
