"""PCIe-inclusive rate of the drop-in call: ParallelSelfPlayWorker.execute_episodes() returning the reference's
list of (state, pi, z) numpy tuples on the host.  Not bench.py's `value`."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import othello_reinforcement_learning_test_amd as pkg

games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(42)
net = pkg.OthelloResNet(10, 128).eval()
w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=50, temperature_threshold=15,
                               num_parallel_games=games, verbose=False)
np.random.seed(0)
w.execute_episodes(max(8, games // 8))          # warm-up
t0 = time.time()
data = w.execute_episodes(games)
dt = time.time() - t0
t1 = time.time()
st, pi, z = w._run_device(games, True)          # device play + one bulk copy, no tuple list
dt_dev = time.time() - t1
print("execute_episodes(%d): %.2f s = %.1f games/s with the tuple list on the host (%d tuples, lanes %d); "
      "same call without building the list: %.2f s = %.1f games/s"
      % (games, dt, games / dt, len(data), w.lanes, dt_dev, games / dt_dev))
