import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import othello_reinforcement_learning_test_amd as pkg
torch.manual_seed(42)
net = pkg.OthelloResNet(10, 128).eval()
N = 4096
rng = np.random.Generator(np.random.PCG64(0))
s = rng.integers(0, 2**63, N, dtype=np.int64) & rng.integers(0, 2**63, N, dtype=np.int64)
o = rng.integers(0, 2**63, N, dtype=np.int64) & ~s
ds, do = torch.from_numpy(s).cuda(), torch.from_numpy(o).cuda()
lg = pkg.DeviceBoards.legal_moves(ds, do)
x = pkg.DeviceBoards.tensor_input(ds, do)
netd = net.cuda()
with torch.no_grad(): rl, rv = netd(x)
for prec in sys.argv[1:] or ['f16x3', 'f16', 'f32']:
    ev = pkg.HipResNetEvaluator(net.cpu(), precision=prec)
    logp, v = ev.forward_bits(ds, do, lg)
    torch.cuda.synchronize()
    e1 = (logp - rl).abs().max().item(); e2 = (v - rv).abs().max().item()
    reps = 3 if prec == 'f32' else 10
    t0 = time.time()
    for _ in range(reps): ev.forward_bits(ds, do, lg)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    fl = N * 378.03e6
    print("%s: maxerr logp %.3e v %.3e | %.3f ms / %d pos | %.1f TFLOP/s algorithmic | %.0f pos/s" % (prec, e1, e2, dt*1e3, N, fl/dt/1e12, N/dt), flush=True)
