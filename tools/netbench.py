"""Network-kernel micro-benchmark: ms per launch of N positions for every trunk build, with the max error vs
torch fp32 on the same weights.  usage: python tools/netbench.py [--n 4096] [--nets 10x128x8:f16x3,5x64x8:f32,...]"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench  # noqa: E402  (mflop_per_position)
import othello_reinforcement_learning_test_amd as pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--nets", default="10x128x8:f16x3,10x128x8:f16,10x128x8:f32,5x64x8:f16x3,5x64x8:f32,2x32x8:f16x3,"
                                  "2x32x8:f32,2x16x8:f32,5x64x6:f16x3,5x64x6:f32,2x32x6:f16x3,2x16x6:f32,3x128x6:f16x3,3x128x6:f32")
args = ap.parse_args()
N = args.n
rng = np.random.Generator(np.random.PCG64(0))
for spec in args.nets.split(","):
    dims, prec = spec.split(":")
    nb, nf, bs = (int(t) for t in dims.split("x"))
    torch.manual_seed(42)
    net = pkg.OthelloResNet(nb, nf, board_size=bs).eval()
    occ = rng.random((N, bs, bs)) < 0.5
    own = occ & (rng.random((N, bs, bs)) < 0.5)
    x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((N, bs, bs)) < 0.4)], 1).astype(np.float32)).cuda()
    with torch.no_grad():
        rl, rv = net.cuda()(x)
    ev = pkg.HipResNetEvaluator(net.cpu(), precision=prec)
    logp, v = ev.forward_planes(x)
    torch.cuda.synchronize()
    e1 = (logp - rl).abs().max().item(); e2 = (v - rv).abs().max().item()
    w = (np.uint64(1) << np.arange(bs * bs, dtype=np.uint64))
    xs = x.cpu().numpy().reshape(N, 3, -1).astype(np.uint64)
    bits = [torch.from_numpy(((xs[:, k] * w).sum(1, dtype=np.uint64)).view(np.int64)).cuda() for k in range(3)]
    reps = 10
    ev.forward_bits(*bits)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        ev.forward_bits(*bits, rescue=False)   # back-to-back asynchronous launches (the rescue check synchronises)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    mf = bench.mflop_per_position(nb, nf) * (bs * bs / 64.0)   # conv work scales with the cells (heads are negligible)
    print("%-12s %-6s maxerr logp %.2e v %.2e | %8.3f ms / %d pos | %7.1f TFLOP/s algorithmic | %9.0f pos/s"
          % (dims, prec, e1, e2, dt * 1e3, N, N * mf * 1e6 / dt / 1e12, N / dt), flush=True)
