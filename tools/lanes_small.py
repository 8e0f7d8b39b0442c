"""Do more lanes help engines that do NOT fill the chip?  ParallelSelfPlayWorker (10x128, 15 sims/move) with G slots in 1 / 2 / 4
lanes, `execute_episodes_arrays`-style batch runs of 4 G games: games/s (device side, tuples left on the device).
usage (GPU box, repo root): python tools/lanes_small.py [G ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import othello_reinforcement_learning_test_amd as pkg  # noqa: E402

torch.manual_seed(42)
net = pkg.OthelloResNet(10, 128).eval()
for G in ([int(a) for a in sys.argv[1:]] or [128, 512, 1024, 2048]):
    for lanes in (1, 2, 4):
        if G // lanes < 16:
            continue
        w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=15, num_parallel_games=G, lanes=lanes,
                                       device_slots=G, verbose=False)
        np.random.seed(1)
        w._run_device(G, True)                      # warm-up (kernels, weights)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 4 * G
        st, pi, z = w._run_device(n, True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("G=%5d lanes=%d: %8.1f games/s (%d games, %d tuples, %.2f s)" % (G, lanes, n / dt, n, len(z), dt), flush=True)
        del w
