"""Replay-tuple handling next to the hot path.

* ``augment_symmetries``      -- the 8-fold transform of ``OthelloBitboard.get_symmetries``
  (/root/reference/src/cython/bitboard.pyx:338-370) over whole arrays of tuples on the GPU
  (C ABI ``oth_augment_symmetries``).  The reference's ``augment_data_with_symmetries``
  (self_play.py:166-212) returns its input unchanged; this is the transform it describes.
* ``DeviceReplayBuffer``      -- ``ReplayBuffer`` (/root/reference/src/train/buffer.py:18-136) with the
  storage on the device: FIFO ring of ``(state, pi, z)`` rows, uniform sampling without replacement,
  same exceptions and return shapes, so tuples never round-trip through host lists.  Pure torch
  tensor plumbing (works on CPU tensors too, which is how the CPU tests exercise it).
* ``train_step`` / ``train_epochs`` -- the trainer's optimisation step (trainer.py:243-367: policy cross-entropy on
  log-probabilities + value MSE, optional AMP) fed straight from the device ring, so a sampled minibatch never
  becomes a numpy array.
* ``load_checkpoint_model``   -- reads a trainer checkpoint (trainer.py:375-384) and rebuilds the
  network, inferring blocks/filters from the key names as players.py:186-202 does.
"""
import numpy as np

from . import _lib


def _board_size_of(states):
    bs = int(states.shape[-1])
    if states.dim() != 4 or states.shape[1] != 3 or states.shape[2] != bs or bs not in (6, 8):
        raise ValueError("states must be (n, 3, S, S) with S = 8 or 6, got %s" % (tuple(states.shape),))
    return bs


def augment_symmetries(states, pis, zs):
    """CUDA tensors (n,3,S,S), (n,S*S+1), (n,) -> (8n,3,S,S), (8n,S*S+1), (8n,) for S = 8 (the reference's board) or
    6 (BASELINE configs[4]): out[8*i + k] is symmetry k of sample i, in ``get_symmetries`` order."""
    import torch
    _lib.require_device()
    states, pis, zs = states.contiguous(), pis.contiguous(), zs.contiguous()
    if not (states.is_cuda and pis.is_cuda and zs.is_cuda):
        raise ValueError("augment_symmetries expects CUDA tensors")
    bs = _board_size_of(states)
    npol = bs * bs + 1
    n = int(zs.shape[0])
    if tuple(pis.shape) != (n, npol) or states.shape[0] != n:
        raise ValueError("pis must be (%d, %d) for a %dx%d board, got %s" % (n, npol, bs, bs, tuple(pis.shape)))
    so = torch.empty((8 * n, 3, bs, bs), dtype=torch.float32, device=states.device)
    po = torch.empty((8 * n, npol), dtype=torch.float32, device=states.device)
    zo = torch.empty((8 * n,), dtype=torch.float32, device=states.device)
    _lib.call("oth_augment_symmetries_n", bs, states.data_ptr(), pis.data_ptr(), zs.data_ptr(), n,
              so.data_ptr(), po.data_ptr(), zo.data_ptr(), _lib.current_stream())
    return so, po, zo


def augment_training_data(training_data):
    """List-of-tuples form for a caller that holds the reference's data format: 8x the samples."""
    import torch
    if not training_data:
        return []
    st = torch.from_numpy(np.stack([d[0] for d in training_data])).cuda()
    pi = torch.from_numpy(np.stack([d[1] for d in training_data])).cuda()
    z = torch.tensor([d[2] for d in training_data], dtype=torch.float32).cuda()
    so, po, zo = (t.cpu().numpy() for t in augment_symmetries(st, pi, z))
    return [(so[i].copy(), po[i].copy(), float(zo[i])) for i in range(len(zo))]


class DeviceReplayBuffer:
    """buffer.py:18-136 with device-resident storage.

    ``add`` accepts either the reference's list of ``(state, pi, z)`` tuples or a triple of tensors
    ``(states, pis, zs)`` (e.g. ``SearchEngine.selfplay_device_tensors()``: no host copy).
    Eviction is FIFO like ``deque(maxlen)`` (buffer.py:33); ``sample`` draws uniformly without
    replacement like ``random.sample`` (buffer.py:78) and raises ``ValueError`` when the buffer holds fewer
    than ``batch_size`` items (buffer.py:72-75).
    """

    def __init__(self, max_size=100000, device="cuda", board_size=8):
        import torch
        if board_size not in (6, 8):
            raise ValueError("board_size must be 8 or 6")
        self.max_size = int(max_size)
        self.device = torch.device(device)
        self.board_size = int(board_size)      # 8: the reference's game; 6: BASELINE configs[4] tuples (n,3,6,6)/(n,37)
        self.npol = self.board_size * self.board_size + 1
        self.states = torch.zeros((self.max_size, 3, self.board_size, self.board_size), dtype=torch.float32,
                                  device=self.device)
        self.policies = torch.zeros((self.max_size, self.npol), dtype=torch.float32, device=self.device)
        self.values = torch.zeros((self.max_size,), dtype=torch.float32, device=self.device)
        self._head = 0      # next write position
        self._size = 0

    def __len__(self):
        return self._size

    def is_ready(self, min_size):  # buffer.py:103
        return self._size >= min_size

    def clear(self):
        self._head = 0
        self._size = 0

    def add_single(self, state, policy, value):  # buffer.py:48-57
        self.add([(np.asarray(state, dtype=np.float32), np.asarray(policy, dtype=np.float32), float(value))])

    def add(self, training_data):
        import torch
        if isinstance(training_data, (list,)):
            if not training_data:
                return
            st = torch.from_numpy(np.stack([d[0] for d in training_data]).astype(np.float32))
            pi = torch.from_numpy(np.stack([d[1] for d in training_data]).astype(np.float32))
            z = torch.tensor([float(d[2]) for d in training_data], dtype=torch.float32)
        else:
            st, pi, z = training_data
        z = z.to(self.device, torch.float32).reshape(-1)
        n = int(z.shape[0])
        bs = self.board_size
        if st.numel() != n * 3 * bs * bs or pi.numel() != n * self.npol:   # never reinterpret another board's rows
            raise ValueError("tuples of %d samples do not have the (3,%d,%d) / (%d,) shapes of this buffer: %s, %s"
                             % (n, bs, bs, self.npol, tuple(st.shape), tuple(pi.shape)))
        st = st.to(self.device, torch.float32).reshape(n, 3, bs, bs)
        pi = pi.to(self.device, torch.float32).reshape(n, self.npol)
        if n >= self.max_size:  # only the newest max_size items survive, oldest first
            st, pi, z = st[n - self.max_size:], pi[n - self.max_size:], z[n - self.max_size:]
            self.states.copy_(st); self.policies.copy_(pi); self.values.copy_(z)
            self._head, self._size = 0, self.max_size
            return
        idx = (torch.arange(n, device=self.device) + self._head) % self.max_size
        self.states[idx] = st
        self.policies[idx] = pi
        self.values[idx] = z
        self._head = (self._head + n) % self.max_size
        self._size = min(self.max_size, self._size + n)

    def _logical_index(self, k):
        """Ring position of the k-th oldest item."""
        start = (self._head - self._size) % self.max_size
        return (start + k) % self.max_size

    def sample(self, batch_size):
        """-> (states (B,3,S,S), policies (B,S*S+1), values (B,1)) tensors on the buffer's device."""
        import torch
        if self._size < batch_size:   # buffer.py:72-75
            raise ValueError(f"Buffer size ({self._size}) is smaller than batch size ({batch_size})")
        pick = torch.randperm(self._size, device=self.device)[:batch_size]   # uniform, without replacement (buffer.py:78)
        if self.device.type != "cuda":   # CPU tensors (the host-side tests): plain indexing
            idx = self._logical_index(pick)
            return self.states[idx], self.policies[idx], self.values[idx].reshape(-1, 1)
        st = torch.empty((batch_size, 3, self.board_size, self.board_size), dtype=torch.float32, device=self.device)
        pi = torch.empty((batch_size, self.npol), dtype=torch.float32, device=self.device)
        v = torch.empty((batch_size,), dtype=torch.float32, device=self.device)
        start = (self._head - self._size) % self.max_size
        _lib.call("oth_replay_gather_n", self.board_size, self.states.data_ptr(), self.policies.data_ptr(), self.values.data_ptr(),
                  pick.data_ptr(), int(batch_size), int(start), int(self.max_size), st.data_ptr(), pi.data_ptr(),
                  v.data_ptr(), _lib.current_stream())
        return st, pi, v.reshape(-1, 1)

    def ordered(self):
        """All items oldest-first (for tests / checkpoints)."""
        import torch
        idx = self._logical_index(torch.arange(self._size, device=self.device))
        return self.states[idx], self.policies[idx], self.values[idx]

    def get_statistics(self):  # buffer.py:107-136: same keys (trainer.py:210-211 reads value_mean / value_std)
        if self._size == 0:
            return {"size": 0, "max_size": self.max_size, "fill_rate": 0.0, "value_mean": 0.0, "value_std": 0.0}
        _, _, v = self.ordered()
        return {
            "size": self._size, "max_size": self.max_size, "fill_rate": self._size / self.max_size,
            "value_mean": float(v.mean()), "value_std": float(v.std(unbiased=False)),   # np.std: population std
            "value_min": float(v.min()), "value_max": float(v.max()),                   # extras
        }


def policy_loss(policy_logits, target_policies):
    """trainer.py:330-346: -mean(sum(target * log_probs, dim=1)) (the network already outputs log-softmax)."""
    import torch
    return -torch.mean(torch.sum(target_policies * policy_logits, dim=1))


def value_loss(value_pred, target_values):
    """trainer.py:348-363: MSE."""
    import torch
    return torch.nn.functional.mse_loss(value_pred, target_values)


def train_step(model, optimizer, states, target_policies, target_values, scaler=None):
    """trainer.py:283-328: one optimisation step on a minibatch of tensors already on the model's device;
    AMP when a GradScaler is given and the tensors are on the GPU.  Returns the total loss as a float."""
    import torch
    optimizer.zero_grad()
    if scaler is not None and states.is_cuda:
        with torch.amp.autocast("cuda"):
            logp, v = model(states)
            total = policy_loss(logp, target_policies) + value_loss(v, target_values)
        scaler.scale(total).backward()
        scaler.step(optimizer)
        scaler.update()
    else:
        logp, v = model(states)
        total = policy_loss(logp, target_policies) + value_loss(v, target_values)
        total.backward()
        optimizer.step()
    return total.item()


def train_epochs(model, optimizer, buffer, num_epochs, batch_size, scaler=None):
    """trainer.py:243-280 with the minibatches gathered on the device (no numpy round trip): model.train(), one
    sampled minibatch per epoch, mean loss.  The model must live on the buffer's device."""
    model.train()
    total, batches = 0.0, 0
    for _ in range(num_epochs):
        states, target_policies, target_values = buffer.sample(batch_size)
        total += train_step(model, optimizer, states, target_policies, target_values, scaler)
        batches += 1
    return total / batches


def infer_architecture(state_dict):
    """(num_blocks, num_filters, board_size) from a state_dict.  Blocks and filters from the parameter names and the
    stem's shape as players.py:186-202 does; the board size from the policy head's FC (net.py:81: S*S+1 rows, so
    65 -> 8 and 37 -> 6), which the reference ignores -- its from_checkpoint always builds an 8x8 net and a 6x6
    checkpoint (configs/debug_6x6.yaml) fails in load_state_dict there."""
    blocks = 0
    for k in state_dict:
        if k.startswith("res_blocks."):
            blocks = max(blocks, int(k.split(".")[1]) + 1)
    filters = int(state_dict["conv_block.conv.weight"].shape[0])
    rows = int(state_dict["policy_head.fc.weight"].shape[0])
    board = int(round((rows - 1) ** 0.5))
    if board * board + 1 != rows or int(state_dict["policy_head.fc.weight"].shape[1]) != 2 * board * board:
        raise ValueError("policy_head.fc.weight %s is not (S*S+1, 2*S*S) for any board size S"
                         % (tuple(state_dict["policy_head.fc.weight"].shape),))
    return blocks, filters, board


def load_checkpoint_model(path, map_location="cpu"):
    """Trainer checkpoint (trainer.py:375-384: dict with 'model_state_dict') or a bare state_dict ->
    ``net.OthelloResNet`` in eval mode.  Loaded with ``weights_only=True`` (nothing from the file executes)."""
    import torch

    from .net import OthelloResNet
    obj = torch.load(path, map_location=map_location, weights_only=True)
    sd = obj["model_state_dict"] if isinstance(obj, dict) and "model_state_dict" in obj else obj
    blocks, filters, board = infer_architecture(sd)
    model = OthelloResNet(blocks, filters, board_size=board).eval()
    model.load_state_dict(sd)
    return model
