#!/usr/bin/env python3
"""bench.py -- self-play games/sec (8x8, 50 MCTS sims/move) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one process
per GPU with torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env).  W untimed
warmup steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize(), MAX over
ranks; rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one batch of synthetic input: every rank plays
`--waves` x `--games` (default 3 x 4096) complete self-play games through `--games` = 4096 concurrent
game slots (BASELINE.json configs[1]; finished slots are refilled; the slots are driven as `--lanes` = 2
independent groups on two streams so that one group's trunk launches fill the other's tails and tree phases)
from the initial position -- 10-block x 128-filter network with seeded-random weights (torch.manual_seed(42)), 50
simulations per move, c_puct 1.0, temperature threshold 15 -- entirely on the device, and (N>1)
the ranks all-gather the replay tuples over RCCL.  value = games completed by all ranks / time.

Extra objects on the same line:
  roofline     dominant kernel = the fused ResNet trunk (k_trunk): algorithmic FLOPs (378.03 MFLOP per
               evaluated position) / its summed launch time, measured live with HIP events on
               the launch stream, against the dense fp16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline the CPU oracle (a C port of the reference algorithm, oracle/) timed on the host cores
               over a bounded sample of the same workload.  Reported, not targeted.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; read when the HIP runtime starts

MFLOP_PER_POSITION = 378.03     # 10x128 network, SURVEY.md 8(d) / BASELINE.md section 3


def mflop_per_position(blocks, filters):
    """Algorithmic MFLOP of one forward pass (2 x MACs): stem 3x3x3 -> F, 2*blocks 3x3 F -> F convs on 64 cells, the two
    1x1 head convs and the three head FCs.  10x128: 189 014 400 MACs = 378.03 MFLOP."""
    f = filters
    macs = 64 * 27 * f + 2 * blocks * 64 * 9 * f * f + 64 * f * 2 + 64 * f + 128 * 65 + 64 * 256 + 256
    return 2.0 * macs / 1e6
PEAK_F16_TFLOPS = 2500.0        # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PLIES_PER_GAME = 61.0           # BASELINE.md work model (used only to scale the CPU sample)


def cpu_baseline(net, sims, budget_s):
    """Time the oracle (kind 'port') on the host cores: one independent serial self-play stream per
    core, bounded to a few plies each, scaled to games/s with the 61-plies-per-game work model."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    onet = ol.Net(net.num_blocks, net.num_filters, ol.state_dict_blob(net.state_dict()))
    cfg = ol.SelfplayCfg(sims, 15, 1, 1.0, 0.3, 0.25, 1, 0)

    def run(plies):
        ev, th = C.c_int64(0), C.c_int(0)
        t0 = time.time()
        n = ol.lib().orc_cpu_baseline(onet.h, C.byref(cfg), cores, plies, 42, C.byref(ev), C.byref(th))
        return n, ev.value, th.value, time.time() - t0
    n, ev, th, dt = run(1)                       # calibration: one ply per stream
    plies = max(1, min(20, int(budget_s / max(dt, 1e-3))))
    if plies > 1:
        n, ev, th, dt = run(plies)
    return {
        "value": round((n / PLIES_PER_GAME) / dt, 5), "unit": "games/s", "cores": th, "kind": "port",
        "sample": "%d plies (%d network evals) over %d independent games, first %d plies each, "
                  "10x128 net fp32, %d sims/move, %.1f s; scaled with 61 plies/game"
                  % (n, ev, cores, plies, sims, dt),
        "evals_per_s": round(ev / dt, 1),
    }


def proportional_shares(rates, nominal, lanes):
    """Games per rank for the next step: proportional to the measured rates, within +-10 % of `nominal`, every share a
    multiple of `lanes`, job total exactly len(rates) * nominal (the rounding remainder goes to the fastest rank)."""
    import numpy as np
    rate = np.asarray(rates, dtype=np.float64)
    world = len(rate)
    if not np.all(np.isfinite(rate)) or rate.min() <= 0:
        return [nominal] * world
    want = np.clip(nominal * world * rate / rate.sum(), 0.9 * nominal, 1.1 * nominal)
    new = [int(w) // lanes * lanes for w in want]
    new[int(np.argmax(rate))] += nominal * world - sum(new)
    return new


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games (= games per step) per GPU")
    ap.add_argument("--waves", type=int, default=3,
                    help="games per step per GPU = waves x --games, played through --games slots with refill")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--precision", default=None, help="f16x3 (default for 128 filters), f16, f32")
    ap.add_argument("--lanes", type=int, default=2,
                    help="independent game groups per GPU, each --games/--lanes slots on its own stream and host "
                         "thread (their kernels overlap: tails and tree phases of one lane are filled by the other)")
    ap.add_argument("--eval-cache", type=int, default=0,
                    help="log2 entries of the opt-in evaluation cache (0 = off; the headline number is measured with it OFF)")
    ap.add_argument("--equal-shares", action="store_true",
                    help="N>1: give every rank exactly --waves x --games per step instead of rate-proportional shares")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the baseline")
    args = ap.parse_args()

    import numpy as np
    import torch

    import othello_reinforcement_learning_test_amd as pkg
    from othello_reinforcement_learning_test_amd import distributed as D

    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    pkg._lib.require_device()   # no GPU => fail loudly
    import torch.distributed as dist

    torch.manual_seed(42)
    net = pkg.OthelloResNet(args.blocks, args.filters).eval()
    ev = pkg.HipResNetEvaluator(net, precision=args.precision)
    import threading
    lanes = max(1, args.lanes)
    assert args.games % lanes == 0
    engs = [pkg.SearchEngine(args.games // lanes, args.sims, temperature_threshold=15, c_puct=1.0, evaluator=ev,
                             eval_cache_log2=args.eval_cache) for _ in range(lanes)]
    for e_ in engs:
        e_.set_timing(True)
    eng = engs[0]
    streams = [torch.cuda.Stream(device=torch.cuda.current_device()) for _ in range(lanes)] if lanes > 1 else [None]

    # OTHELLO_FORCE_DIST=1 under torchrun --nproc-per-node 1: run the RCCL calls of the N>1 path on a one-rank group
    use_dist = world > 1 or (dist.is_available() and dist.is_initialized())
    gloo = use_dist and dist.get_backend() == "gloo"   # rehearsal mode: collectives on host copies

    dev = torch.cuda.current_device()

    def barrier():
        if use_dist:
            if gloo:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev])
        torch.cuda.synchronize()

    # Games per step: world x waves x games in total.  MI355X devices sustain clocks several per cent apart on this
    # MFMA-dense work, and the ranks meet at every step's all-gather, so equal shares would run the job at the pace of
    # its slowest GPU.  Each rank's share of the NEXT step follows its measured rate in the step before (shares are
    # computed identically on every rank from one tiny all-gather, kept within +-10 % of equal, total unchanged).
    nominal = args.games * args.waves
    shares = [nominal] * world
    played = [0]   # games played by the whole job so far in the timed region (filled by step())

    def rebalance(my_games, my_seconds):
        if world == 1 or args.equal_shares:
            return
        t = torch.tensor([my_games / max(my_seconds, 1e-6)], dtype=torch.float64, device="cpu" if gloo else "cuda")
        allr = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allr, t)
        shares[:] = proportional_shares(allr.cpu().numpy(), nominal, lanes)

    lane_errors = []

    def run_lane(k, i, n_lane):
        try:
            torch.cuda.set_device(dev)   # the HIP current device is per thread and new threads start on device 0
            seed = 42 + 1000003 * ((i * world + rank) * lanes + k)
            if streams[k] is None:
                engs[k].selfplay_run(n_lane, seed, add_noise=True)
            else:
                with torch.cuda.stream(streams[k]):
                    engs[k].selfplay_run(n_lane, seed, add_noise=True)
        except BaseException as exc:   # a failing lane must fail the whole bench, not leave stale tuples behind
            lane_errors.append(exc)

    def step(i):
        mine = shares[rank]
        played[0] += sum(shares)
        t_play = time.time()
        if lanes == 1:
            run_lane(0, i, mine)
            if lane_errors:
                raise lane_errors[0]
            st, pi, z = eng.selfplay_device_tensors()
        else:
            ths = [threading.Thread(target=run_lane, args=(k, i, mine // lanes)) for k in range(lanes)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
            if lane_errors:
                raise lane_errors[0]
            parts = [e_.selfplay_device_tensors() for e_ in engs]
            st, pi, z = (torch.cat([p_[j] for p_ in parts]) for j in range(3))
        t_play = time.time() - t_play
        if use_dist:   # the one exchange step: RCCL all-gather of the replay tuples
            if gloo:
                st, pi, z = st.cpu(), pi.cpu(), z.cpu()
            st, pi, z, _ = D.all_gather_replay(st, pi, z, force=True)
        rebalance(mine, t_play)
        return int(z.shape[0])

    barrier()   # also creates the RCCL communicator outside the timed region (matters when --warmup 0)
    for i in range(args.warmup):
        step(i)
    barrier()
    played[0] = 0
    t0 = time.time()
    samples = 0
    stats = {"evals": 0, "simulations": 0, "plies": 0, "games": 0, "net_batches": 0, "terminal_sims": 0,
             "cache_hits": 0}
    kt = {"net_ms": 0.0, "net_launches": 0, "tree_ms": 0.0, "tree_launches": 0}
    union_ms = 0.0   # time during which at least one trunk launch was running (lanes overlap)
    last_shares = list(shares)
    for i in range(args.steps):
        last_shares = list(shares)
        samples = step(args.warmup + i)
        spans = []
        for e_ in engs:
            for k, v in e_.counters().items():
                stats[k] += v
            for k, v in e_.kernel_time().items():
                kt[k] += v
            spans.append(e_.net_spans())
        sp = np.concatenate(spans)
        sp = sp[np.argsort(sp[:, 0])]
        cur_s, cur_e = sp[0]
        for s_, e2 in sp[1:]:
            if s_ > cur_e:
                union_ms += cur_e - cur_s
                cur_s, cur_e = s_, e2
            else:
                cur_e = max(cur_e, e2)
        union_ms += cur_e - cur_s
    barrier()
    dt = time.time() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_games = played[0]
    assert total_games == args.games * args.waves * args.steps * world

    if rank == 0:
        traffic = None   # HBM-side bytes per k_trunk launch from the committed PMC profile (4096 positions)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_trunk_traffic.json")) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"]
        except Exception:
            pass
        # With one lane the union equals the sum of the launch durations; with several lanes the launches of the
        # lanes overlap on the device, so FLOPs are divided by the time during which the kernel was running at all.
        net_s = union_ms * 1e-3
        flops = stats["evals"] * mflop_per_position(args.blocks, args.filters) * 1e6
        achieved = flops / net_s / 1e12 if net_s > 0 else 0.0
        prec = ev.precision
        # MFMA FLOPs the trunk issues per algorithmic FLOP: 3 products of the fp16x3 split, minus the tiles whose
        # source row is zero padding (1/12 of the conv work is skipped by the shipped kernel)
        issued = (3.0 if prec == "f16x3" else 1.0) * (11.0 / 12.0 if prec != "f32" else 1.0)
        out = {
            "metric": "self-play games/sec (8x8, 50 MCTS sims/move)",
            "value": round(total_games / dt, 3),
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f16x3": "f16x3 (fp16 hi/lo operand split on MFMA, fp32 accumulate; fp32-equivalent)",
                      "f16": "f16 (single fp16 MFMA pass, fp32 accumulate)", "f32": "f32"}[prec],
            "data": "synthetic",
            "config": {
                "workload": "8x8, %d sims/move, %d-block x %d ResNet, %d concurrent games on 1 MI355X per rank"
                            % (args.sims, args.blocks, args.filters, args.games),
                "games_per_step_per_gpu": args.games * args.waves, "concurrent_games_per_gpu": args.games, "weights": "seeded random init (torch.manual_seed(42)), eval mode",
                "c_puct": 1.0, "temperature_threshold": 15, "dirichlet": "alpha 0.3 eps 0.25 (no effect on this search)",
                "parallelism": "dp%d: games sharded, %s" % (world, "RCCL all-gather of replay tuples per step"
                                                            if world > 1 else "single GPU"),
                "lanes_per_gpu": lanes,
                "games_per_rank_last_step": ("equal" if world == 1 or args.equal_shares else
                                             "proportional to each rank's measured rate in the previous step "
                                             "(+-10 %% of equal, job total fixed): %s" % last_shares),
                "eval_cache": ("off (every position the search reaches is evaluated by the network)" if not args.eval_cache
                               else "ON: 2^%d entries, %d hits -- NOT the headline configuration" % (args.eval_cache, stats["cache_hits"])),
                "samples_last_step": samples,
                "evals_per_game": round(stats["evals"] / max(1, stats["games"]), 1),
                "plies_per_game": round(stats["plies"] / max(1, stats["games"]), 2),
            },
            "roofline": {
                "kernel": "k_trunk (fused ResNet forward)", "bound": "mfma",
                "achieved": round(achieved, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                "traffic_basis": "PMC FETCH_SIZE/WRITE_SIZE of a full launch of 4096 positions "
                                 "(profiles/r01_trunk_traffic.json); algorithmic bytes of that launch: 1.18 MB",
                "frac_mfma_issue": round(achieved * issued / PEAK_F16_TFLOPS, 4),
                "launches": kt["net_launches"],
                "avg_launch_ms": round(kt["net_ms"] / max(1, kt["net_launches"]), 4),
                "busy_ms": round(union_ms, 1), "concurrent_lanes": lanes,
                "time_basis": "union of the HIP-event intervals of all k_trunk launches (the %d lanes' launches "
                              "overlap; sum of launch durations = %.0f ms)" % (lanes, kt["net_ms"]),
                "positions_per_launch": round(stats["evals"] / max(1, kt["net_launches"]), 1),
                "mfma_flops_issued_per_algorithmic_flop": round(issued, 3),
                "net_time_share": round(net_s / (dt / max(1, 1)) if dt > 0 else 0.0, 4),
                "tree_kernels_ms": round(kt["tree_ms"], 2),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(net, args.sims, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
