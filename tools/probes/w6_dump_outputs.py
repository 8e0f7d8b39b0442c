"""Outputs of the 6x6 5x64 trunk (or DUMP_NET=BxFxS, DUMP_PREC=f32|f16x3|...) of whatever library OTHELLO_MI355X_LIB selects, on fixed inputs -> an .npz; with two arguments:
compare two such files bit for bit.  usage: w6_dump_outputs.py OUT.npz   |   w6_dump_outputs.py A.npz B.npz"""
import sys

import numpy as np

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    same = all(np.array_equal(a[k], b[k]) for k in ("logp", "v"))
    print("outputs of %s vs %s: %s (max |dlogp| %.3e, max |dv| %.3e)"
          % (sys.argv[1], sys.argv[2], "BIT-IDENTICAL" if same else "DIFFERENT",
             np.abs(a["logp"] - b["logp"]).max(), np.abs(a["v"] - b["v"]).max()))
    sys.exit(0 if same else 1)
import torch

sys.path.insert(0, '.')
import othello_reinforcement_learning_test_amd as pkg  # noqa: E402

import os
nb, nf, bs = (int(t) for t in os.environ.get("DUMP_NET", "5x64x6").split("x"))   # DUMP_NET=10x128x8: another network
N = 4099             # (a ragged tail: 4099 = 512 workgroups of eight + three positions)
rng = np.random.Generator(np.random.PCG64(7))
torch.manual_seed(42)
net = pkg.OthelloResNet(nb, nf, board_size=bs).eval()
occ = rng.random((N, bs, bs)) < 0.6
own = occ & (rng.random((N, bs, bs)) < 0.5)
x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((N, bs, bs)) < 0.4)], 1).astype(np.float32)).cuda()
ev = pkg.HipResNetEvaluator(net, precision=os.environ.get("DUMP_PREC") or None)
logp, v = ev.forward_planes(x)
with torch.no_grad():
    rl, rv = net.cuda()(x)
print("%s: kernel %s, max |dlogp| vs torch fp32 %.2e, |dv| %.2e" % (pkg._lib.LIB_PATH.split("/")[-2], ev.kernel_info(N)["kernel"].split(" ")[0],
                                                                  (logp - rl).abs().max().item(), (v - rv).abs().max().item()))
np.savez(sys.argv[1], logp=logp.cpu().numpy(), v=v.cpu().numpy())
