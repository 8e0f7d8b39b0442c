"""CPU experiment: error of a 1-D Winograd trunk (F(2,3) or F(4,3) along the board's x axis, the three row taps
stay direct) with the fp16x3 operand split and fp32 accumulation, against float64 -- next to the direct fp16x3
trunk the shipped kernel implements.  The transformed weights G.g are computed in float64 on the host and then split
into two f16; the input transform B^T.d runs in fp32 on the post-ReLU activations, the result is split into two f16.
usage: python tools/winograd1d_numerics.py [blocks]   (CPU only)"""
import copy
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from winograd_numerics import conv_direct, fold, heads, mm3, trained_like  # noqa: E402

MATS = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                      [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
                     dtype=torch.float64)),
}


def conv_wino1d(x, w64, mode, m):
    """x: n,c,8,8 (dtype of the mode); w64: o,c,3,3 float64.  Tiles of m outputs along x, m+2 inputs."""
    BT, G, AT = MATS[m]
    n, c = x.shape[:2]
    a = m + 2
    U = torch.einsum("ai,ocyi->yaoc", G, w64)                 # 3, a, O, C  (float64 on the host)
    xp = F.pad(x, (1, 1, 1, 1))                               # n,c,10,10
    tiles = xp.unfold(3, a, m)                                # n,c,10,8/m,a
    V = torch.einsum("ai,ncyti->ayntc", BT.to(x.dtype), tiles)  # a,10,n,T,c
    out = None
    for dy in range(3):
        Vd = V[:, dy:dy + 8]                                  # a,8,n,T,c : input row y+dy-1 (padded index y+dy)
        Vd = Vd.reshape(a, -1, c)
        Ud = U[dy].transpose(1, 2)                            # a, c, o
        if mode == "f64":
            M = Vd @ Ud
        else:
            M = torch.stack([mm3(Vd[k].contiguous(), Ud[k].float().contiguous()) for k in range(a)])
        out = M if out is None else out + M
    M = out.reshape(a, 8, n, 8 // m, -1)                      # a,y,n,T,o
    Y = torch.einsum("ia,aynto->noyti", AT.to(x.dtype), M)    # n,o,y,T,m
    return Y.reshape(n, -1, 8, 8)


def trunk(net, x, conv, mode):
    dt = torch.float64 if mode == "f64" else torch.float32
    w, b = fold(net.conv_block.conv, net.conv_block.bn)
    h = F.relu(conv_direct(x.to(dt), w.to(dt), "f64" if mode == "f64" else "f32") + b.to(dt)[None, :, None, None])
    for blk in net.res_blocks:
        w1, b1 = fold(blk.conv1, blk.bn1)
        w2, b2 = fold(blk.conv2, blk.bn2)
        y = F.relu(conv(h, w1, mode) + b1.to(dt)[None, :, None, None])
        y = conv(y, w2, mode) + b2.to(dt)[None, :, None, None]
        h = F.relu(y + h)
    return h


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    net = trained_like(blocks)
    rng = np.random.default_rng(0)
    n = 96
    own = rng.random((n, 8, 8)) < 0.3
    opp = (rng.random((n, 8, 8)) < 0.4) & ~own
    leg = (rng.random((n, 8, 8)) < 0.15) & ~own & ~opp
    x = torch.tensor(np.stack([own, opp, leg], 1).astype(np.float32))
    net64 = copy.deepcopy(net).double()
    with torch.no_grad():
        ref_l, ref_v = net64(x.double())
        t32_l, t32_v = net(x)
        print("torch fp32 vs f64: dlogp %.2e dv %.2e" % ((t32_l.double() - ref_l).abs().max(), (t32_v.double() - ref_v).abs().max()))
        direct = lambda h, w, mode: conv_direct(h, w.to(h.dtype), mode)  # noqa: E731
        h64 = trunk(net, x, direct, "f64")
        for name, conv, mode in (("direct f16x3", direct, "x3"),
                                 ("wino F(2,3) f64", lambda h, w, md: conv_wino1d(h, w, md, 2), "f64"),
                                 ("wino F(2,3) f16x3", lambda h, w, md: conv_wino1d(h, w, md, 2), "x3"),
                                 ("wino F(4,3) f64", lambda h, w, md: conv_wino1d(h, w, md, 4), "f64"),
                                 ("wino F(4,3) f16x3", lambda h, w, md: conv_wino1d(h, w, md, 4), "x3")):
            h = trunk(net, x, conv, mode)
            l, v = heads(net64, h.double())
            print("%-18s trunk relerr %.2e   dlogp vs f64 %.2e  vs torch-fp32 %.2e   dv vs f64 %.2e" %
                  (name, ((h.double() - h64).abs().max() / h64.abs().max()), (l - ref_l).abs().max(),
                   (l - t32_l.double()).abs().max(), (v - ref_v).abs().max()))


main()
