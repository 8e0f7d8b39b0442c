"""CPU emulation behind the run-time activation scale of the fp16-split trunks: how much precision does
the hi/lo operand split lose when the activations are carried x 8, x 4, x 2 or x 1 instead of x 16?

The kernels split every conv operand into two f16, a = a_hi + a_lo, and accumulate three products in fp32.  Activations
are PRE-SCALED by a power of two s before the split so that a_lo = f16(s*a - f16(s*a)) -- 2^-11 of a_hi and smaller --
stays clear of the f16 subnormals (spacing 2^-24): at s = 16 an activation of 0.01 has a_lo ~ 8e-5 with 10 significant
bits left; at s = 1 it has 6.  A saturated launch is rescued by halving s (range 1875 -> 30 000 in the Winograd trunks), so
the question is what the smaller s costs.  Emulated here: the direct 3x3 trunk and the 1-D Winograd F(2,3) trunk (input
transform in fp32, then the split), weights x their power-of-two scale (|w| <= 16384: never near a subnormal), fp32
accumulation, on the trained-like 6- and 10-block 128-filter networks of tests/test_gpu_parity.py, against float64 --
and on the same networks with the stem scaled up by 512 (activations in the thousands: the case that needs the rescue),
their heads scaled down by the same factor so that the outputs stay comparable.

usage: python tools/act_scale_numerics.py [blocks ...]      (CPU only; ~1 minute per network)"""
import copy
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, "."); sys.path.insert(0, "tools")
import winograd_numerics as wn  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
SCALE = [16.0]


def pow2_scale(w):   # the packers' per-layer weight scale: the largest power of two with max |w| * scale <= 16384
    mx = float(w.abs().max())
    return 2.0 ** np.floor(np.log2(16384.0 / mx)) if mx > 0 else 1.0


def mm3_scaled(a, b):
    """a: activations (fp32, natural units), b: weights (fp32).  The kernel's arithmetic: operands (s*a) and (ws*b) split
    into two f16 (torch rounds to f16 subnormals exactly as the hardware conversion does), three products, fp32
    accumulation; the epilogue multiplies by 1 / ws and the heads by 1 / s."""
    s, ws = SCALE[0], pow2_scale(b)
    ah, al = wn.split(a * s)
    bh, bl = wn.split(b * ws)
    return (ah @ bh + (ah @ bl + al @ bh)) / (s * ws)


def conv_wino1d(x, w64, mode):
    n, c = x.shape[:2]
    U = torch.einsum("ai,ocyi->yaoc", G, w64)
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(3, 4, 2)
    V = torch.einsum("ai,ncyti->ayntc", BT.to(x.dtype), tiles)   # the input transform, in fp32 (as the epilogue does it)
    out = None
    for dy in range(3):
        Vd = V[:, dy:dy + 8].reshape(4, -1, c)
        Ud = U[dy].transpose(1, 2)
        M = Vd @ Ud if mode == "f64" else torch.stack([wn.mm3(Vd[k].contiguous(), Ud[k].float().contiguous()) for k in range(4)])
        out = M if out is None else out + M
    M = out.reshape(4, 8, n, 4, -1)
    return torch.einsum("ia,aynto->noyti", AT.to(x.dtype), M).reshape(n, -1, 8, 8)


def trunk(net, x, conv, mode):
    dt = torch.float64 if mode == "f64" else torch.float32
    w, b = wn.fold(net.conv_block.conv, net.conv_block.bn)
    h = F.relu(wn.conv_direct(x.to(dt), w.to(dt), "f64" if mode == "f64" else "f32") + b.to(dt)[None, :, None, None])
    for blk in net.res_blocks:
        w1, b1 = wn.fold(blk.conv1, blk.bn1)
        w2, b2 = wn.fold(blk.conv2, blk.bn2)
        y = F.relu(conv(h, w1, mode) + b1.to(dt)[None, :, None, None])
        y = conv(y, w2, mode) + b2.to(dt)[None, :, None, None]
        h = F.relu(y + h)
    return h


def run(blocks, boost):
    net = wn.trained_like(blocks)
    if boost != 1.0:   # activations x boost through the whole trunk, outputs unchanged (up to the betas of the blocks)
        with torch.no_grad():
            net.conv_block.bn.weight.mul_(boost)
            net.conv_block.bn.bias.mul_(boost)
            net.policy_head.conv.weight.div_(boost)
            net.value_head.conv.weight.div_(boost)
    rng = np.random.default_rng(0)
    n = 96
    own = rng.random((n, 8, 8)) < 0.3
    opp = (rng.random((n, 8, 8)) < 0.4) & ~own
    leg = (rng.random((n, 8, 8)) < 0.15) & ~own & ~opp
    x = torch.tensor(np.stack([own, opp, leg], 1).astype(np.float32))
    net64 = copy.deepcopy(net).double()
    direct = lambda h, w, mode: wn.conv_direct(h, w.to(h.dtype), mode)  # noqa: E731
    with torch.no_grad():
        ref_l, ref_v = net64(x.double())
        t32_l, t32_v = net(x)
        h64 = trunk(net, x, direct, "f64")
        amax = float(h64.abs().max())
        print("== %d x 128, stem x %g: largest activation %.0f; torch fp32 vs float64: dlogp %.2e dv %.2e"
              % (blocks, boost, amax, (t32_l.double() - ref_l).abs().max(), (t32_v.double() - ref_v).abs().max()))
        wn.mm3 = mm3_scaled
        for s in (16.0, 8.0, 4.0, 2.0, 1.0):
            SCALE[0] = s
            row = "   act scale %4g (clamp: Winograd %6.0f, direct %6.0f)" % (s, 30000 / s, 60000 / s)
            for name, conv, lim in (("direct", direct, 60000 / s), ("F(2,3)", conv_wino1d, 30000 / s)):
                if amax > lim:
                    row += " | %s: SATURATES" % name
                    continue
                h = trunk(net, x, conv, "x3")
                lp, v = wn.heads(net64, h.double())
                row += " | %s: dlogp vs f64 %.2e, vs torch fp32 %.2e, dv %.2e" % (
                    name, (lp - ref_l).abs().max(), (lp - t32_l.double()).abs().max(), (v - ref_v).abs().max())
            print(row, flush=True)


if __name__ == "__main__":
    for b in ([int(a) for a in sys.argv[1:]] or [6, 10]):
        run(b, 1.0)
        run(b, 512.0)
