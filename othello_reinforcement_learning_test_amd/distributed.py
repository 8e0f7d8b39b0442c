"""Multi-GPU self-play: one process per GPU, games sharded across ranks, ONE exchange step.

Self-play games are independent units (no cross-game state, weights read-only while playing), so the
path shards with no data-path collective: rank r plays its share of the episodes with its own RNG
stream.  The only exchange is the end-of-iteration all-gather of the replay tuples
``(state f32[n,3,S,S], pi f32[n,S*S+1], z f32[n])`` (S = 8, or 6 for BASELINE configs[4]) so that every (replicated) trainer feeds the same
``ReplayBuffer`` -- the multi-GPU form of ``replay_buffer.add(training_data)``
(/root/reference/src/train/trainer.py:185; the reference itself is single-process).

Backend: ``nccl`` (= RCCL over xGMI on ROCm) for CUDA tensors, ``gloo`` for the CPU tests.  Tuple
counts differ per rank (games have 57-66+ plies), so counts are gathered first and the payload is
gathered padded to the largest count (one all_gather_into_tensor per array: three large collectives,
~63 KB per game) and trimmed afterwards.
"""
import numpy as np


def shard_episodes(num_episodes, rank, world_size):
    """Round-robin share of rank `rank`: episodes rank, rank+world, ...."""
    if world_size <= 1:
        return int(num_episodes)
    return int((num_episodes - rank + world_size - 1) // world_size) if num_episodes > rank else 0


def force_dist_from_env():
    """OTHELLO_FORCE_DIST: run the collectives even on a one-rank group (the RCCL smoke test of the GPU suite).  Unset, empty,
    "0", "false", "no" and "off" mean OFF (`bool("0")` used to turn it on)."""
    import os
    return os.environ.get("OTHELLO_FORCE_DIST", "0").strip().lower() not in ("", "0", "false", "no", "off")


_GATHER_BUFFERS = {}   # (device, dtype, row shape, world) -> [rows, padded input, gathered, compacted output]


def _gather_buffers(dev, dtype, row, world, nmax, total):
    """Persistent buffers of the exchange: after the first step the exchange itself allocates nothing (a caller that asks
    DistributedSelfPlayWorker for owned tensors -- copy=True, its default -- pays one clone of the result per call on top:
    0.83 GB at 8 ranks x 100 k tuples; only copy=False is allocation-free end to end).  Capacity grows by powers of
    two; at 8 ranks x 100 k tuples the three arrays hold 8 x 100 k x (768 + 260 + 4) B = 0.83 GB gathered plus the
    same compacted, 0.10 GB padded input -- 1.8 GB of the 288 GB, allocated once."""
    import torch
    key = (str(dev), dtype, tuple(row), world)
    buf = _GATHER_BUFFERS.get(key)
    if buf is None or buf[0] < nmax:
        rows = 1024
        while rows < nmax:
            rows *= 2
        buf = [rows, torch.zeros((rows,) + tuple(row), dtype=dtype, device=dev),
               torch.empty((world * rows,) + tuple(row), dtype=dtype, device=dev),
               torch.empty((world * rows,) + tuple(row), dtype=dtype, device=dev)]
        _GATHER_BUFFERS[key] = buf
    return buf


def release_gather_buffers():
    """Drop the persistent exchange buffers (up to ~1.8 GB at 8 ranks x 100 k tuples): call when self-play is over and
    the memory is wanted for something else; the next exchange allocates them again.  Views returned by earlier
    ``all_gather_replay`` calls keep their storage alive until they are dropped too."""
    _GATHER_BUFFERS.clear()


def all_gather_replay(states, pis, zs, group=None, force=False):
    """All-gather variable-length replay tuples.  Inputs are torch tensors (or lists of tensors, concatenated in
    order) on one device (CUDA for RCCL, CPU for gloo) with a common leading length n_r.  Returns (states, pis, zs, counts) where the
    arrays are the concatenation over ranks in rank order and counts[r] = n_r.

    The returned arrays are VIEWS of persistent buffers (valid until the next call on this device): the padded input
    and the gathered output are reused from step to step, equal counts return the gathered buffer itself, unequal
    counts are compacted with world slice copies into a second persistent buffer -- no per-step allocation here, no
    torch.cat (a caller that needs to keep the result across the next call clones it: execute_episodes_tensors(copy=True))."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        if isinstance(zs, (list, tuple)):
            states, pis, zs = (torch.cat(list(t)) for t in (states, pis, zs))
        return states, pis, zs, [int(zs.shape[0])]   # force=True runs the collectives on a one-rank group (RCCL smoke test)
    world = dist.get_world_size(group)
    # each argument may also be a LIST of tensors (e.g. one per engine lane): the parts are written one after the other
    # straight into the persistent padded buffer, so the caller needs no torch.cat of its own
    states, pis, zs = ([t] if not isinstance(t, (list, tuple)) else list(t) for t in (states, pis, zs))
    dev = zs[0].device
    mine = sum(int(t.shape[0]) for t in zs)
    n = torch.tensor([mine], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.cpu().tolist()
    nmax, total = max(counts), sum(counts)
    out = []
    for parts in (states, pis, zs):
        row = tuple(parts[0].shape[1:])
        if nmax == 0:
            out.append(parts[0][:0])
            continue
        _, pad, gathered, compact = _gather_buffers(dev, parts[0].dtype, row, world, nmax, total)
        pad_v = pad[:nmax]
        at = 0
        for t in parts:
            pad_v[at: at + t.shape[0]] = t
            at += int(t.shape[0])
        gath_v = gathered[: world * nmax]
        dist.all_gather_into_tensor(gath_v, pad_v, group=group)
        if all(c == nmax for c in counts):
            out.append(gath_v)
            continue
        off = 0
        for r in range(world):
            compact[off: off + counts[r]] = gath_v[r * nmax: r * nmax + counts[r]]
            off += counts[r]
        out.append(compact[:total])
    return out[0], out[1], out[2], counts


def broadcast_model(model, src=0, group=None):
    """Rank `src`'s weights to every rank -- the once-per-iteration step of a replicated trainer (every rank
    trains the same replicated model on the same all-gathered tuples, or only rank `src` trains and the others follow).  All
    floating parameters and buffers travel as ONE flattened tensor (11.9 MB fp32 for 10x128: a single collective, not 144),
    the int64 `num_batches_tracked` counters as a second one; tensors are written back in place, so the evaluator's
    `refresh()` sees the new version counters.  A no-op outside an initialised process group or at world size 1.  The
    reference is single-process (/root/reference/src/train/trainer.py:180-185 trains the model the worker reads)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    tensors = list(model.parameters()) + list(model.buffers())
    moved = 0
    for is_float in (True, False):
        part = [t for t in tensors if t.is_floating_point() == is_float]
        if not part:
            continue
        dev = part[0].device
        if dist.get_backend(group) == "gloo":
            dev = torch.device("cpu")
        flat = torch.cat([t.detach().reshape(-1).to(device=dev, dtype=part[0].dtype) for t in part])
        dist.broadcast(flat, src=src, group=group)
        at = 0
        with torch.no_grad():
            for t in part:
                n = t.numel()
                t.copy_(flat[at: at + n].view_as(t))
                at += n
        moved += flat.numel() * flat.element_size()
    return moved


class DistributedSelfPlayWorker:
    """``execute_episodes`` with the reference's signature for a job of WORLD_SIZE processes: each rank
    plays ``shard_episodes`` games on its own GPU, then all ranks receive all tuples."""

    def __init__(self, worker, rank=None, world_size=None, base_seed=0, group=None, force_collectives=None):
        import os

        import torch.distributed as dist
        self.worker = worker  # a ParallelSelfPlayWorker bound to this rank's GPU
        self.group = group
        # run the collectives even on a one-rank group (the RCCL smoke test of the GPU suite: OTHELLO_FORCE_DIST=1)
        self.force_collectives = force_dist_from_env() if force_collectives is None else bool(force_collectives)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self.base_seed = int(base_seed)
        self._calls = 0

    def execute_episodes_tensors(self, num_episodes, add_dirichlet_noise=True, copy=True):
        """-> CUDA tensors (states, pis, zs) of the WHOLE job, plus per-rank tuple counts.

        ``copy=True`` (default): fresh tensors the caller owns -- a trainer may keep the result of call k alive while it
        makes call k+1 (the reference's replay_buffer.add keeps references, trainer.py:185).  ``copy=False``: zero-copy
        VIEWS of the persistent exchange buffers (``all_gather_replay``) or of the engine's compacted tuples, valid only
        until the next call on this device -- for consumers that copy at once (bench.py, DeviceReplayBuffer.add)."""
        mine = shard_episodes(num_episodes, self.rank, self.world_size)
        if getattr(self.worker, "_streaming", False):
            raise RuntimeError("DistributedSelfPlayWorker needs a batch-mode worker (continuous=False): a batch run "
                               "would drop the stream's in-flight games")
        if hasattr(self.worker, "_grow_engine"):
            self.worker._grow_engine(mine)   # auto slot width: the whole share at once (results do not depend on it)
        eng = self.worker.engine
        self.worker.batch_mcts.evaluator.refresh()
        seed = (self.base_seed + 0x9E3779B97F4A7C15 * (self._calls * self.world_size + self.rank + 1)) % 2**63
        self._calls += 1
        import torch
        if mine > 0:
            eng.selfplay_run_rescued(mine, seed, add_dirichlet_noise)   # (restarted from its seed if a launch saturated)
            st, pi, z = eng.selfplay_device_tensors()
        else:
            st = torch.empty((0, 3, eng.board_size, eng.board_size), dtype=torch.float32, device="cuda")   # 6x6: (0,3,6,6)
            pi = torch.empty((0, eng.npol), dtype=torch.float32, device="cuda")
            z = torch.empty((0,), dtype=torch.float32, device="cuda")
        st, pi, z, counts = all_gather_replay(st, pi, z, self.group, force=self.force_collectives)
        if copy:
            st, pi, z = st.clone(), pi.clone(), z.clone()
        return st, pi, z, counts

    def execute_episodes(self, num_episodes, add_dirichlet_noise=True):
        st, pi, z, _ = self.execute_episodes_tensors(num_episodes, add_dirichlet_noise, copy=False)
        st, pi, z = st.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        return [(st[i].copy(), pi[i].copy(), float(z[i])) for i in range(len(z))]


def init_from_env(backend=None):
    """torch.distributed init from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  Single process when WORLD_SIZE is absent or 1."""
    import os

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev:
        torch.cuda.set_device(local % ndev)   # one GPU per rank on a real node; shared only in rehearsals
    force = force_dist_from_env() and "RANK" in os.environ   # one-rank RCCL smoke test
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # OTHELLO_DIST_BACKEND=gloo: rehearse the N>1 control flow where RCCL cannot run
            # (CPU-only container, or several ranks sharing one GPU)
            backend = os.environ.get("OTHELLO_DIST_BACKEND") or ("nccl" if ndev else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
