"""CPU experiment: error of a Winograd F(2x2,3x3) trunk with the fp16x3 operand split (fp32 accumulation)
against float64, next to the direct-convolution fp16x3 trunk the shipped kernel implements.  Decides whether
a Winograd kernel can stay inside the 1e-4 tolerance on trained-like weights."""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from othello_reinforcement_learning_test_amd.net import OthelloResNet


def trained_like(blocks, seed=123):
    torch.manual_seed(seed)
    net = OthelloResNet(blocks, 128).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 1.2 + 0.1)
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 1.8 + 0.3)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(torch.exp(torch.randn(mod.weight.shape[0], 1, 1, 1, generator=g) * 0.45))
        net.policy_head.fc.weight.mul_(2.0)
    return net


def fold(conv, bn):
    w = conv.weight.double()
    s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    return (w * s[:, None, None, None]), (bn.bias.double() - bn.running_mean.double() * s)


def split(x):
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi.float(), lo.float()


def mm3(a, b):  # fp16x3 product, fp32 accumulation
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh)


def conv_direct(x, w, mode):
    n = x.shape[0]
    cols = F.unfold(x, 3, padding=1)                      # n, C*9, 64
    a = cols.transpose(1, 2).reshape(n * 64, -1)
    b = w.reshape(w.shape[0], -1).t()
    if mode == "f64":
        y = a @ b
    elif mode == "f32":
        y = a @ b
    else:
        y = mm3(a, b.contiguous())
    return y.reshape(n, 64, -1).transpose(1, 2).reshape(n, -1, 8, 8)


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def conv_wino(x, w64, mode):
    n, c = x.shape[:2]
    U = torch.einsum("ai,ocij,bj->aboc", G, w64, G)          # 4,4,O,C  (float64 on the host)
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                 # n,c,4,4,4,4 (ty,tx,i,j)
    bt = BT.to(x.dtype)
    V = torch.einsum("ai,nctuij,bj->abntuc", bt, tiles, bt)   # 4,4,n,4,4,c   in x.dtype
    V = V.reshape(16, n * 16, c)
    U = U.reshape(16, U.shape[2], c).transpose(1, 2)          # 16, c, o
    if mode == "f64":
        M = V @ U
    else:
        U = U.float()
        M = torch.stack([mm3(V[k], U[k].contiguous()) for k in range(16)])
    M = M.reshape(4, 4, n, 4, 4, -1)
    at = AT.to(x.dtype)
    Y = torch.einsum("ia,abntuo,jb->notiuj", at, M, at)       # n,o,ty,i,tx,j
    return Y.reshape(n, -1, 8, 8)


def trunk(net, x, conv, mode):
    dt = torch.float64 if mode == "f64" else torch.float32
    w, b = fold(net.conv_block.conv, net.conv_block.bn)
    # first layer (3 input planes) stays a direct convolution in every variant
    h = F.relu(conv_direct(x.to(dt), w.to(dt), "f64" if mode == "f64" else "f32") + b.to(dt)[None, :, None, None])
    for blk in net.res_blocks:
        w1, b1 = fold(blk.conv1, blk.bn1)
        w2, b2 = fold(blk.conv2, blk.bn2)
        if conv is conv_wino:
            y = F.relu(conv(h, w1, mode) + b1.to(dt)[None, :, None, None])
            y = conv(y, w2, mode) + b2.to(dt)[None, :, None, None]
        else:
            y = F.relu(conv(h, w1.to(dt), mode) + b1.to(dt)[None, :, None, None])
            y = conv(y, w2.to(dt), mode) + b2.to(dt)[None, :, None, None]
        h = F.relu(y + h)
    return h


def heads(net, h):
    net64 = net
    p = net64.policy_head(h)
    v = net64.value_head(h)
    return p, v


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    net = trained_like(blocks)
    rng = np.random.default_rng(0)
    n = 96
    own = rng.random((n, 8, 8)) < 0.3
    opp = (rng.random((n, 8, 8)) < 0.4) & ~own
    leg = (rng.random((n, 8, 8)) < 0.15) & ~own & ~opp
    x = torch.tensor(np.stack([own, opp, leg], 1).astype(np.float32))
    import copy
    net64 = copy.deepcopy(net).double()
    with torch.no_grad():
        ref_l, ref_v = net64(x.double())
        t32_l, t32_v = net(x)
        print("torch fp32 vs f64: dlogp %.2e dv %.2e" % ((t32_l.double() - ref_l).abs().max(), (t32_v.double() - ref_v).abs().max()))
        h64 = trunk(net, x, conv_direct, "f64")
        l, v = heads(net64, h64)
        print("folded f64 trunk vs torch f64: %.2e" % (l - ref_l).abs().max())
        for name, conv, mode in (("direct f32", conv_direct, "f32"), ("direct f16x3", conv_direct, "x3"),
                                 ("wino f64", conv_wino, "f64"), ("wino f16x3", conv_wino, "x3")):
            h = trunk(net, x, conv, mode)
            l, v = heads(net64, h.double())
            print("%-14s trunk relerr %.2e   dlogp vs f64 %.2e  vs torch-fp32 %.2e   dv vs f64 %.2e  max|V-range|" %
                  (name, ((h.double() - h64).abs().max() / h64.abs().max()), (l - ref_l).abs().max(),
                   (l - t32_l.double()).abs().max(), (v - ref_v).abs().max()))
        print("activation max %.1f" % h64.abs().max())


if __name__ == "__main__":
    main()
