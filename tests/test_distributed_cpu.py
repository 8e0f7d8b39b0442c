"""N>1 path on CPU: world_size-2 gloo processes exercise the episode sharding and the variable-length
replay all-gather (the only exchange step of the multi-GPU path; RCCL on the GPU box)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from othello_reinforcement_learning_test_amd.distributed import all_gather_replay, shard_episodes


def test_shard_episodes_partitions_exactly():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 5, 7, 8, 100, 4096, 4099):
            shares = [shard_episodes(n, r, world) for r in range(world)]
            assert sum(shares) == n and max(shares) - min(shares) <= 1
            assert shares == sorted(shares, reverse=True)   # round-robin: low ranks get the remainder


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(rank, n):
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    st = torch.from_numpy((rng.random((n, 3, 8, 8)) < 0.3).astype(np.float32))
    pi = torch.from_numpy(rng.random((n, 65)).astype(np.float32))
    z = torch.from_numpy(rng.integers(-1, 2, n).astype(np.float32))
    return st, pi, z


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st, pi, z = _make(rank, counts[rank])
        gs, gp, gz, got = all_gather_replay(st, pi, z)
        ok = got == list(counts)
        exp = [_make(r, counts[r]) for r in range(world)]
        ok &= torch.equal(gs, torch.cat([e[0] for e in exp]))
        ok &= torch.equal(gp, torch.cat([e[1] for e in exp]))
        ok &= torch.equal(gz, torch.cat([e[2] for e in exp]))
        q.put((rank, bool(ok), int(gz.shape[0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(61, 64), (120, 0), (1, 300)])
def test_all_gather_replay_gloo_world2(counts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, n in res:
        assert ok and n == sum(counts), (rank, ok, n)


def test_single_process_is_identity():
    st, pi, z = _make(0, 10)
    a, b, c, counts = all_gather_replay(st, pi, z)
    assert a is st and b is pi and c is z and counts == [10]


def test_bench_proportional_shares():
    """bench.py's N>1 load balancing: shares follow the measured rates, stay within 10 % of equal, are multiples of the
    lane count and keep the job total fixed; degenerate inputs fall back to equal shares."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    nominal, lanes = 12288, 2
    rates = [500.0, 520.0, 480.0, 510.0, 495.0, 530.0, 470.0, 505.0]
    s = bench.proportional_shares(rates, nominal, lanes)
    assert sum(s) == nominal * 8 and all(x % lanes == 0 for x in s)
    assert all(0.9 * nominal <= x <= 1.1 * nominal + lanes * 8 for x in s)
    order = sorted(range(8), key=lambda i: rates[i])
    assert [s[i] for i in order] == sorted(s)                       # faster rank, more games
    finish = [s[i] / rates[i] for i in range(8)]
    assert max(finish) / min(finish) < 1.002                        # everybody finishes together
    assert max(nominal / r for r in rates) / (sum(finish) / 8) > 1.05   # equal shares would have cost > 5 %
    assert bench.proportional_shares([1.0, 0.0], nominal, lanes) == [nominal, nominal]
    assert bench.proportional_shares([1.0, float("nan")], nominal, lanes) == [nominal, nominal]
    s = bench.proportional_shares([100.0, 300.0], nominal, lanes)   # clamp at +-10 %
    assert sum(s) == 2 * nominal and min(s) >= int(0.9 * nominal) - lanes and max(s) <= 1.1 * nominal + lanes
    # one fast rank among slow ones: nobody leaves the +-10 % band (the rest is re-divided among the others)
    for rates2 in ([100.0] * 7 + [300.0], [300.0] * 7 + [100.0], [1.0, 1.02, 0.97, 5.0], [3006.0, 966.0, 949.0, 4000.0]):
        for nom, ln in ((1536, 2), (64, 2), (12288, 2), (2048, 1)):
            s2 = bench.proportional_shares(rates2, nom, ln)
            assert sum(s2) == nom * len(rates2) and all(x % ln == 0 for x in s2)
            assert min(s2) >= 0.9 * nom - ln and max(s2) <= 1.1 * nom + ln, (rates2, nom, s2)


def _worker_steps(rank, world, port, q):
    """bench.py's exchange step at world size `world` on CPU tensors: several steps through the SAME persistent
    buffers (growing, shrinking, equal and empty counts; two lane parts per rank), then the rate all-gather +
    proportional_shares every rank must evaluate identically."""
    import importlib.util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location(
            "bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        ok = True
        plans = [[40 + 3 * r for r in range(world)], [7] * world, [0 if r % 3 == 0 else 90 - r for r in range(world)],
                 [1500 if r == world - 1 else 2 for r in range(world)], [0] * world, [33] * world]
        for step, counts in enumerate(plans):
            st, pi, z = _make(100 * step + rank, counts[rank])
            cut = counts[rank] // 3   # two "lanes"
            gs, gp, gz, got = all_gather_replay([st[:cut], st[cut:]], [pi[:cut], pi[cut:]], [z[:cut], z[cut:]])
            exp = [_make(100 * step + r, counts[r]) for r in range(world)]
            ok &= got == counts and int(gz.shape[0]) == sum(counts)
            ok &= torch.equal(gs, torch.cat([e[0] for e in exp])) and torch.equal(gp, torch.cat([e[1] for e in exp]))
            ok &= torch.equal(gz, torch.cat([e[2] for e in exp]))
        rate = torch.tensor([500.0 + 7.0 * rank], dtype=torch.float64)
        allr = torch.zeros(world, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, rate)
        shares = bench.proportional_shares(allr.numpy(), 1536, 2)
        q.put((rank, bool(ok), shares))
    finally:
        dist.destroy_process_group()


def test_exchange_step_world8_gloo():
    """The N = 8 form of BASELINE configs[2]'s exchange (counts, three padded all-gathers through persistent buffers,
    rate all-gather, shares) by eight gloo ranks on CPU tensors -- the GPU box admits only six processes on its card,
    so this is where world size 8 is exercised."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_steps, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    shares = [s for _, _, s in res]
    assert all(s == shares[0] for s in shares) and sum(shares[0]) == 8 * 1536   # identical on every rank, total fixed


class _FakeEngine:
    """Stands in for SearchEngine in the CPU test below: seeded CPU tuples instead of a device run."""
    board_size, npol = 8, 65

    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def selfplay_run(self, n, seed, noise):
        self.calls += 1
        self._t = _make(1000 * self.calls + self.rank, 50 * n + self.rank)

    selfplay_run_rescued = selfplay_run   # (SearchEngine's form that restarts a run whose network launches saturated)

    def selfplay_device_tensors(self):
        return self._t


class _FakeWorker:
    def __init__(self, rank):
        self.engine = _FakeEngine(rank)
        self.batch_mcts = type("B", (), {"evaluator": type("E", (), {"refresh": staticmethod(lambda: None)})()})()


def _worker_hold(rank, world, port, q):
    """A trainer holding the result of call k across call k+1 (round-3 advisor finding): execute_episodes_tensors
    returns tensors the caller owns by default; copy=False returns views of the persistent exchange buffers, which the
    next call overwrites; release_gather_buffers drops them."""
    from othello_reinforcement_learning_test_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = D.DistributedSelfPlayWorker(_FakeWorker(rank), base_seed=3)
        a = w.execute_episodes_tensors(4)                       # 2 games per rank -> 100 + rank tuples each
        keep = [t.clone() for t in a[:3]]
        b = w.execute_episodes_tensors(4)
        ok = a[3] == [100, 101] and all(torch.equal(x, y) for x, y in zip(a[:3], keep))      # call 1 survived call 2
        ok &= not torch.equal(a[0], b[0])
        c = w.execute_episodes_tensors(4, copy=False)
        cv = [t.clone() for t in c[:3]]
        d = w.execute_episodes_tensors(4, copy=False)
        ok &= c[0].data_ptr() == d[0].data_ptr() and not torch.equal(c[0], cv[0])           # views: overwritten
        exp = [_make(1000 * 4 + r, 100 + r) for r in range(world)]
        ok &= torch.equal(d[0], torch.cat([e[0] for e in exp])) and torch.equal(d[2], torch.cat([e[2] for e in exp]))
        ok &= len(D._GATHER_BUFFERS) == 3
        D.release_gather_buffers()
        ok &= len(D._GATHER_BUFFERS) == 0
        e = w.execute_episodes_tensors(4)                      # allocates again
        ok &= e[3] == [100, 101] and len(D._GATHER_BUFFERS) == 3
        q.put((rank, bool(ok), 0))
    finally:
        dist.destroy_process_group()


def test_distributed_worker_results_survive_the_next_call():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hold, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)


def _worker_bcast(rank, world, port, q):
    from othello_reinforcement_learning_test_amd import OthelloResNet
    from othello_reinforcement_learning_test_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # every rank starts from DIFFERENT weights
        net = OthelloResNet(2, 16)
        with torch.no_grad():
            net.conv_block.bn.num_batches_tracked.fill_(7 + rank)
        v0 = sum(int(t._version) for t in list(net.parameters()) + list(net.buffers()))
        moved = D.broadcast_model(net, src=0)
        torch.manual_seed(100)
        want = OthelloResNet(2, 16)
        ok = all(torch.equal(a, b) for (ka, a), (kb, b) in zip(net.state_dict().items(), want.state_dict().items())
                 if not ka.endswith("num_batches_tracked"))
        ok &= int(net.conv_block.bn.num_batches_tracked) == 7
        ok &= moved == sum(t.numel() * t.element_size() for t in list(net.parameters()) + list(net.buffers()))
        ok &= sum(int(t._version) for t in list(net.parameters()) + list(net.buffers())) > v0   # refresh() would re-upload
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_broadcast_model_gloo_world2():
    """The once-per-iteration weight broadcast of a replicated trainer: rank 0's parameters AND buffers (BatchNorm
    running statistics, the int64 batch counters) reach every rank in two collectives, in place."""
    from othello_reinforcement_learning_test_amd import OthelloResNet
    from othello_reinforcement_learning_test_amd import distributed as D
    assert D.broadcast_model(OthelloResNet(2, 16)) == 0     # no process group: a no-op
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bcast, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
