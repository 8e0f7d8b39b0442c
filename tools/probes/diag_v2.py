import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import othello_reinforcement_learning_test_amd as pkg
N=4096; bs=8
rng = np.random.Generator(np.random.PCG64(0))
torch.manual_seed(42)
net = pkg.OthelloResNet(2, 32, board_size=bs).eval()
occ = rng.random((N, bs, bs)) < 0.5
own = occ & (rng.random((N, bs, bs)) < 0.5)
x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((N, bs, bs)) < 0.4)], 1).astype(np.float32)).cuda()
with torch.no_grad():
    rl, rv = net.cuda()(x)
ev = pkg.HipResNetEvaluator(net.cpu(), precision="f16x3")
l0, v0 = ev.forward_planes(x); torch.cuda.synchronize(); l0=l0.clone(); v0=v0.clone()
for r in range(3):
    l, v = ev.forward_planes(x); torch.cuda.synchronize()
    print("run", r, "logp bitwise equal to run 0:", bool((l == l0).all().item()), "| v differs at", int((v != v0).sum().item()), "positions; v vs ref bad:", int(((v.ravel()-rv.ravel()).abs()>1e-5).sum().item()))
for n in (1024, 1536, 2048, 3072, 4096):
    bad=0
    for r in range(3):
        l, v = ev.forward_planes(x[:n]); torch.cuda.synchronize()
        bad += int(((v.ravel()-rv.ravel()[:n]).abs()>1e-5).sum().item())
    print("n", n, "bad v over 3 runs:", bad)
