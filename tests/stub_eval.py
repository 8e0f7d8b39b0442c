"""Closed-form stub evaluator shared by the golden generator, the oracle tests and the GPU tests.

A position (self, opp) is hashed (splitmix64 finaliser); action a gets the logit
STUB_LOGITS[hash(pos, a) & 15] and the position gets a dyadic value in [-1, 1).  Because priors
come from a 16-entry table whose torch.exp image is stored in tests/golden/g3_search.npz
(``stub_exp``), the search parity tests need no libm / torch.exp agreement: the checker feeds
``stub_exp[idx]`` as the "probabilities" the reference saw.  The values are not normalised on
purpose: the renormalisation over legal moves (reference node.py:71-80) is part of what is pinned.
"""
import numpy as np

U64 = np.uint64
# 16 irregular float32 logits (not dyadic, so f32 summation order matters and is pinned)
STUB_LOGITS = (np.arange(16, dtype=np.float64) * -0.3717 - 0.9137).astype(np.float32)

_C1 = U64(0xBF58476D1CE4E5B9)
_C2 = U64(0x94D049BB133111EB)
_G = U64(0x9E3779B97F4A7C15)
_A = U64(0xD1B54A32D192ED03)


def _mix(x):
    x = x.astype(U64)
    x = x ^ (x >> U64(30))
    x = x * _C1
    x = x ^ (x >> U64(27))
    x = x * _C2
    x = x ^ (x >> U64(31))
    return x


def planes_to_bits(x):
    """(N,3,8,8) float 0/1 planes -> (self u64[N], opp u64[N])."""
    x = np.asarray(x)
    w = (U64(1) << np.arange(64, dtype=U64))
    s = (x[:, 0].reshape(-1, 64) > 0.5).astype(U64) @ w
    o = (x[:, 1].reshape(-1, 64) > 0.5).astype(U64) @ w
    return s.astype(U64), o.astype(U64)


def stub_index_value(self_b, opp_b, npol=65):
    """-> (idx int[N,npol] in 0..15, value f32[N]).  npol = 65 (8x8) or 37 (6x6)."""
    with np.errstate(over="ignore"):
        s = np.asarray(self_b, dtype=U64).reshape(-1)
        o = np.asarray(opp_b, dtype=U64).reshape(-1)
        h = _mix(s ^ _mix(o + _G))
        a = np.arange(npol, dtype=U64)[None, :]
        idx = ((_mix(h[:, None] + a * _A) >> U64(33)) & U64(15)).astype(np.int64)
        v = (((h >> U64(40)) & U64(0xFF)).astype(np.int64) - 128).astype(np.float32) / np.float32(128)
    return idx, v


def stub_logits_values(x, return_index=False):
    """Planes (N,3,8,8) -> (logits f32[N,65], values f32[N])."""
    s, o = planes_to_bits(x)
    idx, v = stub_index_value(s, o)
    logits = STUB_LOGITS[idx]
    if return_index:
        return logits, v, idx
    return logits, v


def stub_probs_values(self_b, opp_b, exp_table, npol=65):
    """(self, opp) -> (probs f32[N,npol] = exp_table[idx], values f32[N]) as the reference saw them."""
    idx, v = stub_index_value(self_b, opp_b, npol)
    return np.asarray(exp_table, dtype=np.float32)[idx], v
