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
vs=[]
for r in range(3):
    logp, v = ev.forward_planes(x)
    torch.cuda.synchronize()
    vs.append(v.clone().cpu().numpy().ravel()); 
    e=np.abs(vs[-1]-rv.cpu().numpy().ravel())
    bad=np.nonzero(e>1e-5)[0]
    print("run",r,"maxerr",e.max(),"n bad",len(bad),"bad idx",bad[:20], "bad%4", np.bincount(bad%4,minlength=4), "logp err", (logp-rl).abs().max().item())
    if len(bad): print("   v", vs[-1][bad[:5]], "ref", rv.cpu().numpy().ravel()[bad[:5]])
for n in (1, 3, 4, 64, 256, 1000):
    logp, v = ev.forward_planes(x[:n]); torch.cuda.synchronize()
    e=(v.ravel()-rv.ravel()[:n]).abs()
    print("n",n,"maxerr v",e.max().item(), "nbad", int((e>1e-5).sum()))
