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
