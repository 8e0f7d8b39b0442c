"""Where and how the value output of k_trunk_h3<32, 8, 1, 4> goes wrong (the eight round-4 builds; their build script is in the history at 274f240): 2x32 network
on 8x8, 4096 positions (two workgroups per CU), six launches; per launch the number of positions whose v differs from
launch 0 / from torch fp32, and over all launches: which wave of the workgroup (position % 4), which workgroup
(first 256 = first on their CU), the size and sign of the error, and whether a wrong value repeats."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import othello_reinforcement_learning_test_amd as pkg   # noqa: E402

N, bs = 4096, 8
rng = np.random.Generator(np.random.PCG64(0))
torch.manual_seed(42)
net = pkg.OthelloResNet(2, 32, board_size=bs).eval()
occ = rng.random((N, bs, bs)) < 0.5
own = occ & (rng.random((N, bs, bs)) < 0.5)
x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((N, bs, bs)) < 0.4)], 1).astype(np.float32)).cuda()
with torch.no_grad():
    rl, rv = net.cuda()(x)
rv = rv.ravel().cpu().numpy()
ev = pkg.HipResNetEvaluator(net.cpu(), precision="f16x3")
print("lib:", os.environ.get("OTHELLO_MI355X_LIB", "(product)"), "|", ev.kernel_info(N)["kernel"])
runs = []
for r in range(6):
    l, v = ev.forward_planes(x)
    torch.cuda.synchronize()
    runs.append((l.cpu().numpy().copy(), v.ravel().cpu().numpy().copy()))
bad_any = np.zeros(N, bool)
for r, (l, v) in enumerate(runs):
    bad = np.abs(v - rv) > 1e-5
    bad_any |= bad
    print("launch %d: logp == launch 0: %s | v != launch 0 at %d | v off torch by > 1e-5 at %d (max %.3g)"
          % (r, bool((l == runs[0][0]).all()), int((v != runs[0][1]).sum()), int(bad.sum()), float(np.abs(v - rv).max())))
idx = np.flatnonzero(bad_any)
print("positions ever wrong: %d" % len(idx))
if len(idx):
    print("  by wave of the workgroup (pos %% 4):", np.bincount(idx % 4, minlength=4).tolist())
    blk = idx // 4
    print("  by workgroup: first 256: %d, 256..511: %d, 512..767: %d, 768..1023: %d"
          % tuple(int(((blk >= a) & (blk < a + 256)).sum()) for a in (0, 256, 512, 768)))
    print("  by XCD (workgroup %% 8):", np.bincount(blk % 8, minlength=8).tolist())
    errs = np.stack([v[idx] - rv[idx] for _, v in runs])
    print("  |error| quantiles over wrong launches:", np.quantile(np.abs(errs[np.abs(errs) > 1e-5]), [0, .25, .5, .75, 1]).round(4).tolist())
    wrong_runs = (np.abs(errs) > 1e-5).sum(0)
    print("  launches (of 6) in which a wrong position was wrong:", np.bincount(wrong_runs, minlength=7).tolist())
    for i in idx[:8]:
        print("   pos %4d (wg %4d wave %d): torch %.6f, launches %s" % (i, i // 4, i % 4, rv[i], " ".join("%.6f" % v[i] for _, v in runs)))
