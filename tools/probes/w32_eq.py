import os, sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import othello_reinforcement_learning_test_amd as pkg
N=1000; rng=np.random.Generator(np.random.PCG64(1)); torch.manual_seed(42)
net=pkg.OthelloResNet(10,128).eval()
occ=rng.random((N,8,8))<0.5; own=occ&(rng.random((N,8,8))<0.5)
x=torch.from_numpy(np.stack([own,occ&~own,(~occ)&(rng.random((N,8,8))<0.4)],1).astype(np.float32)).cuda()
ev=pkg.HipResNetEvaluator(net)
l,v=ev.forward_planes(x); torch.cuda.synchronize()
np.save('/tmp/w32_%s.npy' % os.environ.get('OTH_WINO32','0'), np.concatenate([l.cpu().numpy().ravel(), v.cpu().numpy().ravel()]))
print('saved', os.environ.get('OTH_WINO32','0'), float(l.abs().max()))
