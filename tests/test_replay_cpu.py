"""CPU tests of the host-side pieces next to the hot path: the device replay ring
(run here on CPU tensors) against a deque with the reference's semantics, and the checkpoint loader."""
import collections
import random

import numpy as np
import pytest
import torch

from othello_reinforcement_learning_test_amd.net import OthelloResNet
from othello_reinforcement_learning_test_amd.replay import (DeviceReplayBuffer, infer_architecture,
                                                            load_checkpoint_model)


def _tuples(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [((rng.random((3, 8, 8)) < 0.3).astype(np.float32), rng.random(65).astype(np.float32),
             float(rng.integers(-1, 2))) for _ in range(n)]


def test_ring_matches_deque_fifo():
    buf = DeviceReplayBuffer(max_size=50, device="cpu")
    ref = collections.deque(maxlen=50)            # reference buffer.py:33
    for step, n in enumerate((7, 30, 25, 3, 120, 9)):   # includes wrap-around and an add larger than capacity
        data = _tuples(n, step)
        buf.add(data)
        ref.extend(data)
        assert len(buf) == len(ref)
        st, pi, z = buf.ordered()
        assert np.array_equal(st.numpy(), np.stack([d[0] for d in ref]))
        assert np.array_equal(pi.numpy(), np.stack([d[1] for d in ref]))
        assert np.array_equal(z.numpy(), np.array([d[2] for d in ref], dtype=np.float32))
    # tensor-triple form of add
    st = torch.zeros(4, 3, 8, 8); pi = torch.ones(4, 65); z = torch.tensor([1., -1., 0., 1.])
    buf.add((st, pi, z))
    assert torch.equal(buf.ordered()[2][-4:], z)


def test_sample_shapes_and_errors():
    buf = DeviceReplayBuffer(max_size=100, device="cpu")
    with pytest.raises(ValueError):               # buffer.py:72-75
        buf.sample(1)
    buf.add(_tuples(40, 1))
    assert not buf.is_ready(41) and buf.is_ready(40)
    torch.manual_seed(0)
    s, p, v = buf.sample(32)
    assert tuple(s.shape) == (32, 3, 8, 8) and tuple(p.shape) == (32, 65) and tuple(v.shape) == (32, 1)  # test_train.py:43-59
    # without replacement: 40 of 40 is a permutation
    s, p, v = buf.sample(40)
    allp = buf.ordered()[1]
    assert sorted(map(tuple, p.numpy().round(6).tolist())) == sorted(map(tuple, allp.numpy().round(6).tolist()))
    with pytest.raises(ValueError):
        buf.sample(41)
    st = buf.get_statistics()
    assert st["size"] == 40 and st["capacity"] == 100 and -1 <= st["value_mean"] <= 1
    buf.clear()
    assert len(buf) == 0


def test_checkpoint_loader_roundtrip(tmp_path):
    torch.manual_seed(3)
    net = OthelloResNet(3, 32)
    assert infer_architecture(net.state_dict()) == (3, 32)
    # the trainer's format (reference trainer.py:375-384) and a bare state_dict
    p1, p2 = tmp_path / "ckpt.pt", tmp_path / "sd.pt"
    torch.save({"model_state_dict": net.state_dict(), "global_step": 5, "epoch": 1,
                "config": {"model": {"num_blocks": 3}}}, p1)
    torch.save(net.state_dict(), p2)
    x = torch.rand(2, 3, 8, 8)
    with torch.no_grad():
        ref = net.eval()(x)
    for p in (p1, p2):
        m = load_checkpoint_model(str(p))
        assert (m.num_blocks, m.num_filters) == (3, 32) and not m.training
        with torch.no_grad():
            out = m(x)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
