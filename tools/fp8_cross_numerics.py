"""VERDICT r2 item 1 (optional third experiment), gated on a CPU emulation: the two cross terms of the fp16 operand split,
a_hi*b_lo + a_lo*b_hi, on the block-scaled fp8 MFMA (2x the f16 rate) with a_hi*b_hi staying f16.  Emulated optimistically:
every fp8 operand keeps 4 significant bits with a PERFECT per-element exponent (the real instruction shares one scale per
32 elements, which can only be worse).  Same network, positions and float64 reference as tools/split_numerics.py.
Gate: build only if max |dlogp| < 5e-5.   usage: python tools/fp8_cross_numerics.py   (CPU only)"""
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_lib as ol  # noqa: E402
from othello_reinforcement_learning_test_amd.net import OthelloResNet  # noqa: E402


def f16(t):
    return t.to(torch.float16).to(torch.float64)


def lo16(t):
    return (t - f16(t)).to(torch.float16).to(torch.float64)


def q4(t):   # 4 significant bits (e4m3's 3 stored + the implicit one), ideal exponent
    m, e = torch.frexp(t)
    return torch.ldexp(torch.round(m * 16.0) / 16.0, e)


def forward(net, x, mode):
    conv2d = torch.nn.functional.conv2d

    def conv_bn(conv, bn, a):
        scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        w = conv.weight.double() * scale.view(-1, 1, 1, 1)
        b = bn.bias.double() - bn.running_mean.double() * scale
        p = conv.padding
        if mode == "exact":
            y = conv2d(a, w, padding=p)
        else:
            ah, al, wh, wl = f16(a), lo16(a), f16(w), lo16(w)
            y = conv2d(ah, wh, padding=p)
            if mode == "f16x3":
                y = y + conv2d(ah, wl, padding=p) + conv2d(al, wh, padding=p)
            elif mode == "fp8cross":
                y = y + conv2d(q4(ah), q4(wl), padding=p) + conv2d(q4(al), q4(wh), padding=p)
        return y + b.view(1, -1, 1, 1)
    a = torch.relu(conv_bn(net.conv_block.conv, net.conv_block.bn, x))
    for blk in net.res_blocks:
        r = a
        a = torch.relu(conv_bn(blk.conv1, blk.bn1, a))
        a = torch.relu(conv_bn(blk.conv2, blk.bn2, a) + r)
    ph = net.policy_head
    sc = ph.bn.weight.double() / torch.sqrt(ph.bn.running_var.double() + ph.bn.eps)
    p = torch.relu(conv2d(a, ph.conv.weight.double() * sc.view(-1, 1, 1, 1)) +
                   (ph.bn.bias.double() - ph.bn.running_mean.double() * sc).view(1, -1, 1, 1))
    return torch.log_softmax(p.flatten(1) @ ph.fc.weight.double().t() + ph.fc.bias.double(), dim=1)


torch.manual_seed(123)
net = OthelloResNet(6, 128).eval()
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
rng = np.random.Generator(np.random.PCG64(31))
xs = []
for _ in range(8):
    b = ol.board()
    while not ol.lib().orc_is_terminal(b):
        xs.append(ol.tensor(b))
        mv = ol.legal_list(b)
        ol.lib().orc_make_move(b, int(mv[rng.integers(len(mv))]))
x = torch.from_numpy(np.stack(xs[:200])).double()
with torch.no_grad():
    exact = forward(net, x, "exact")
    for mode in ("f16x3", "fp8cross", "f16x1"):
        lp = forward(net, x, mode)
        print("%-10s max |dlogp| vs float64 = %.2e" % (mode, (lp - exact).abs().max().item()), flush=True)
