"""How many products of the fp16 operand split does the trunk need?  CPU emulation (float64 arithmetic, operands
rounded as the kernel would round them) on the trained-like 6x128 network of tests/test_gpu_parity.py::
test_trunk_on_trained_like_weights, 400 game positions, against a float64 forward.  VERDICT r1 item 8(ii).

  3 products  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi   (shipped: both operands to ~22 bits)
  2 products  (w_hi+w_lo)*x_hi                     (activations rounded to f16, weights split)
  2 products  w_hi*(x_hi+x_lo)                     (weights rounded to f16, activations split)
  1 product   w_hi*x_hi                            (single f16 pass)
usage: python tools/split_numerics.py   (CPU only)"""
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle_lib as ol  # noqa: E402
from othello_reinforcement_learning_test_amd.net import OthelloResNet  # noqa: E402


def f16(t):
    return t.to(torch.float16).to(torch.float64)


def split22(t):   # hi + lo, two f16 (what the kernel carries)
    hi = t.to(torch.float16)
    lo = (t - hi.to(torch.float64)).to(torch.float16)
    return hi.to(torch.float64) + lo.to(torch.float64)


def forward(net, x, wq, xq):
    """float64 forward with conv operands quantised by wq (weights, BN folded) and xq (activations)."""
    def conv_bn(conv, bn, a):
        scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        w = conv.weight.double() * scale.view(-1, 1, 1, 1)
        b = bn.bias.double() - bn.running_mean.double() * scale
        return torch.nn.functional.conv2d(xq(a), wq(w), padding=conv.padding) + b.view(1, -1, 1, 1)
    a = torch.relu(conv_bn(net.conv_block.conv, net.conv_block.bn, x))
    for blk in net.res_blocks:
        r = a
        a = torch.relu(conv_bn(blk.conv1, blk.bn1, a))
        a = torch.relu(conv_bn(blk.conv2, blk.bn2, a) + r)
    ident = lambda t: t  # noqa: E731  (heads run in fp32 VALU in the kernel: keep them exact here)
    p = torch.relu(conv_bn(net.policy_head.conv, net.policy_head.bn, a)) if False else None
    ph, vh = net.policy_head, net.value_head
    sc = ph.bn.weight.double() / torch.sqrt(ph.bn.running_var.double() + ph.bn.eps)
    p = torch.relu(torch.nn.functional.conv2d(a, ph.conv.weight.double() * sc.view(-1, 1, 1, 1)) +
                   (ph.bn.bias.double() - ph.bn.running_mean.double() * sc).view(1, -1, 1, 1))
    logp = torch.log_softmax(p.flatten(1) @ ph.fc.weight.double().t() + ph.fc.bias.double(), dim=1)
    return logp, ident(a)


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
x = torch.from_numpy(np.stack(xs[:400])).double()
with torch.no_grad():
    exact, _ = forward(net, x, lambda t: t, lambda t: t)
    rows = [("3 products (shipped: w and x to ~22 bits)", split22, split22),
            ("2 products: x rounded to f16, w split", split22, f16),
            ("2 products: w rounded to f16, x split", f16, split22),
            ("1 product: single f16 pass", f16, f16)]
    for name, wq, xq in rows:
        lp, _ = forward(net, x, wq, xq)
        print("%-46s max |dlogp| vs float64 = %.2e" % (name, (lp - exact).abs().max().item()))
    lp32, _ = net(x.float())
    print("%-46s max |dlogp| vs float64 = %.2e" % ("torch fp32 forward", (lp32.double() - exact).abs().max().item()))
