#!/usr/bin/env python3
"""bench.py -- self-play games/sec (8x8, 50 MCTS sims/move) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one process
per GPU with torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the env).  W untimed
warmup steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize(), MAX over
ranks; rank 0 prints ONE JSON line on stdout (and a heartbeat line per step on stderr).

A PLAIN `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) launches itself: before torch or the
package is imported -- before anything touches a GPU -- this process starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD (never an exec),
relays its output and exits with its return code.  A WORLD_SIZE that disagrees with --gpus is an error (exit 2), never a
silent single-GPU run that prints "n_gpus": 1.

Workload (BASELINE.json configs[1]): 4096 concurrent 8x8 self-play games per GPU, 50 simulations per move,
10-block x 128-filter network with seeded-random weights (torch.manual_seed(42)), c_puct 1.0, temperature
threshold 15, every position the search reaches evaluated by the network -- entirely on the device.

A "step" = one pass of the hot path over one batch: the 4096 game slots of a rank play at STEADY STATE
(streaming mode: finished slots are refilled at once and games stay in flight between steps, so a step has
no ragged tail) until `--step-games` (default 1536) more games per GPU have finished; their (state, pi, z)
replay tuples are compacted in HBM and (N>1) all-gathered over RCCL.  value = games completed by all ranks
during the timed steps / time.  The slots are driven as `--lanes` = 2 independent groups on two streams so
that one group's trunk launches fill the other's tails and tree phases.

The timed steps run with the engine's HIP-event hooks OFF.  After the timed region one extra PROFILED step
(hooks on, not part of `value`) measures the dominant kernel live for the roofline object:
  roofline     k_trunk (fused ResNet forward): algorithmic FLOPs (378.03 MFLOP per evaluated position,
               SURVEY 8(d)) / the time during which a trunk launch was running (HIP events on the launch
               streams), against the dense fp16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline the CPU oracle (a C port of the reference algorithm, oracle/) timed on the host cores over a
               bounded, phase-uniform sample of the same workload.  Reported, not targeted.
  other_configs (N = 1, default workload only; --no-other-configs skips them) three short legs after the headline, same
               process, same streaming schedule: BASELINE configs[3] (400 sims/move, c_puct 1.5, threshold 20), configs[4]
               (6x6, 25 sims, 5x64 net; rules parity unpinned) and configs[1] with the opt-in evaluation cache -- secondary
               figures; `value`, `config` and `roofline` are the headline's alone.
The roofline object's kernel name and issued-FLOP factor come from the library (oth_net_kernel_info), not from this file.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HSA_ENABLE_IPC_MODE_LEGACY=0 selects dmabuf IPC handles, the only kind this pool's host driver exports (with the legacy
# mode RCCL's cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument`); read when the HIP runtime
# starts.  A DEFAULT only: a caller's own value wins here AND in the self-launched ranks (launch_ranks passes the environment
# on untouched), so an operator can flip it (it has never been exercised with N > 1 ranks).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them used by the default stream): with four lanes two
# of them then share a queue and their launches serialise -- configs[4] on four lanes measured 25.7 k games/s with the default
# and 27.9-28.1 k with 8 (profiles/r05_hw_queues.log; two- and three-lane workloads are unaffected).  Read when the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PEAK_F16_TFLOPS = 2500.0        # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3         # fp32-input MFMA (v_mfma_f32_16x16x4_f32) peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0           # HBM3E peak, MI355X_MICROARCH.md
PLIES_PER_GAME = 60.7           # measured by the engine over >100k games of this workload
# Algorithmic bytes the tree kernel (k_tree, one launch per simulation) moves per game and simulation:
# 24 B leaf position out + 264 B network result in (65 log-probs + value) + ~3 nodes of the descent x (32 B node + 8 edges
# x 16 B) read + the ~3 path edges rewritten by the backup (16 B each) = 816 B.
TREE_BYTES_PER_GAME_SIM = 24 + 264 + 3 * (32 + 8 * 16) + 3 * 16
# Planning figure for the time budget only (tests/test_bench_budget.py): games/s one MI355X sustains on the default
# workload, taken well below the slowest box measured (503-528 in round 1).
PLANNING_RATE = 420.0


def mflop_per_position(blocks, filters, board=8):
    """Algorithmic MFLOP of one forward pass (2 x MACs): stem 3x3x3 -> F, 2*blocks 3x3 F -> F convs on the board's
    cells, the two 1x1 head convs and the three head FCs.  10x128 on 8x8: 189 014 400 MACs = 378.03 MFLOP
    (SURVEY.md 8(d))."""
    f, c = filters, board * board
    macs = c * 27 * f + 2 * blocks * c * 9 * f * f + c * f * 2 + c * f + 2 * c * (c + 1) + c * 256 + 256
    return 2.0 * macs / 1e6


# The secondary legs (`other_configs`), in the order they run, with the wall-clock each took on the box at MEASURED_RATE
# games/s on the headline (round 5: configs[3] with 4416 slots incl. its staggered ramp, configs[4], configs[1] + evaluation
# cache with 8192 slots).  A leg starts only if 1.5 x its (rate-scaled) seconds still fit before LEGS_HARD_STOP seconds of
# process life, so a slow box or a cold start drops legs -- last ones first -- instead of running into the driver's limit.
LEG_SECONDS = (("configs[3]", 67.0), ("configs[4]", 12.0), ("configs[1] + eval cache", 32.0))
MEASURED_RATE = 615.0
OTHER_LEGS_SECONDS = sum(s for _, s in LEG_SECONDS)
LEGS_HARD_STOP = 520.0
DRIVER_LIMIT = 600.0


def planned_seconds(steps, warmup, step_games, slots=4096, stagger=61, profile_steps=1, cpu_budget=20.0,
                    rate=PLANNING_RATE, startup=150.0, legs=True):
    """Wall-clock plan of one bench.py run on one MI355X (default workload): process start-up (the first
    `import torch` on a fresh box can take 2 minutes) + staggered ramp + (warmup + steps + profiled) steps +
    the secondary legs that still fit (each scaled with the rate, admitted by the rule above) + CPU baseline."""
    ramp = 0.5 * slots * min(stagger, 61) / 61.0 / rate          # half-full slots during the staggered start
    t = startup + ramp + (warmup + steps + profile_steps) * step_games / rate
    for _, sec in (LEG_SECONDS if legs else ()):
        sec = sec * MEASURED_RATE / rate
        if t + 1.5 * sec <= LEGS_HARD_STOP:
            t += sec
    # CPU baseline: its budget + the ply in flight when the budget ends, or the 8 plies every stream must play (~5 s each on a
    # loaded 128-core host) if that is longer
    return t + max(1.4 * cpu_budget, 45.0) + 10.0


TRUNK_SOURCES = ("net_wino.hip", "net_epilogue.h", "net_heads.h", "wave_bfly.h", "net.h", "Makefile")


def trunk_source_sha256():
    """Hash of the sources the benchmarked trunk kernel is built from: a committed PMC traffic figure is only this
    kernel's while these files are the ones it was measured with (tools/bench_pmc.sh records the hash)."""
    import hashlib
    h = hashlib.sha256()
    for f in TRUNK_SOURCES:
        with open(os.path.join(ROOT, "othello_reinforcement_learning_test_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def committed_traffic(kernel_name):
    """HBM-side bytes per launch from the newest committed rocprofv3 --pmc passes over bench.py's own launch shape
    (profiles/rNN_bench_traffic.json, written by tools/bench_pmc.sh) -> dict(traffic, tree_traffic, basis, stale, why)."""
    import glob
    out = {"traffic": None, "tree_traffic": None, "basis": None, "stale": None, "why": None}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_traffic.json")))
    if not files:
        return out
    tfile = files[-1]
    try:
        with open(tfile) as f:
            tj = json.load(f)
        tk = tj["kernels"].get("trunk") or tj["kernels"]["k_trunk16"]
        out["traffic"] = tk["traffic_bytes_per_launch"]
        out["tree_traffic"] = tj["kernels"]["k_tree"]["traffic_bytes_per_launch"]
        out["basis"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `%s` (profiles/%s): kernel %s, %.0f positions per "
                        "trunk launch there; gfx950 FETCH_SIZE x2 correction applied to the 16 B/lane weight reads"
                        % (tj["command"], os.path.basename(tfile), tk["kernel"], tk["positions_per_launch"]))
        base = lambda s: s.replace("oth::", "").split("<")[0].split(" ")[0]   # noqa: E731
        why = []
        if base(tk["kernel"]) != base(kernel_name):
            why.append("measured on %s, this run's kernel is %s" % (tk["kernel"], kernel_name.split(" ")[0]))
        sha = tj.get("kernel_source_sha256")
        if sha is None:
            why.append("the file records no source hash")
        elif sha != trunk_source_sha256():
            why.append("the trunk sources changed since it was measured")
        out["stale"], out["why"] = bool(why), "; ".join(why) or None
    except Exception as exc:
        out["why"] = "unreadable: %r" % (exc,)
    return out


PHASE_BIN = 6   # plies per game-phase bin of the CPU baseline's estimator


def seconds_per_game(phase, secs, plies_per_game=PLIES_PER_GAME, bin_plies=PHASE_BIN):
    """Wall seconds ONE stream needs for one game, from sampled plies: mean ply time per game-phase bin (phase = plies played
    before the ply; bins of `bin_plies`) summed over the phases 0 .. plies_per_game of a game.  An empty bin takes the mean of
    its nearest non-empty neighbours.  Weighting by PHASE, not by plies finished: an endgame ply is cheap (many simulations
    end in terminal nodes, no network evaluation), so a stream that started late finishes more plies inside a time budget
    and a plain plies/s average over-weights it."""
    import numpy as np
    phase, secs = np.asarray(phase), np.asarray(secs, dtype=np.float64)
    nb = int(np.ceil(plies_per_game / bin_plies))
    idx = np.minimum(phase // bin_plies, nb - 1)
    tot = np.bincount(idx, weights=secs, minlength=nb)[:nb]
    cnt = np.bincount(idx, minlength=nb)[:nb]
    mean = np.where(cnt > 0, tot / np.maximum(cnt, 1), np.nan)
    if np.isnan(mean).all():
        return float("nan")
    have = np.flatnonzero(~np.isnan(mean))
    for i in np.flatnonzero(np.isnan(mean)):
        lo, hi = have[have < i], have[have > i]
        near = [mean[lo[-1]]] if len(lo) else []
        near += [mean[hi[0]]] if len(hi) else []
        mean[i] = float(np.mean(near))
    # plies of a game in each bin: bin_plies each, the last bin takes what is left of plies_per_game
    w = np.full(nb, float(bin_plies))
    w[-1] = plies_per_game - bin_plies * (nb - 1)
    return float((mean * w).sum())


def phase_weighted_rate(rec_phase, rec_secs, plies, cores, n_boot=2000, seed=1):
    """-> (games/s of `cores` concurrent streams, (lo, hi) bootstrap 95 % interval over STREAMS).  rec_phase / rec_secs:
    [streams, cap] per-ply records, plies[s] of them valid in row s."""
    import numpy as np
    rows = [(rec_phase[s, :plies[s]], rec_secs[s, :plies[s]]) for s in range(len(plies)) if plies[s] > 0]
    est = lambda pick: cores / seconds_per_game(np.concatenate([rows[i][0] for i in pick]),   # noqa: E731
                                                np.concatenate([rows[i][1] for i in pick]))
    value = est(range(len(rows)))
    rng = np.random.Generator(np.random.PCG64(seed))
    boot = np.array([est(rng.integers(0, len(rows), len(rows))) for _ in range(n_boot)])
    boot = boot[np.isfinite(boot)]
    return float(value), (float(np.percentile(boot, 2.5)), float(np.percentile(boot, 97.5)))


def cpu_baseline(net, sims, budget_s, evals_per_game, min_plies=8):
    """Time the oracle (kind 'port') on the host: one serial self-play stream per PHYSICAL core of this process's affinity
    mask, stream s starting 58*s/streams random plies into a game (phase-uniform: openings, middle games and endgames are
    sampled like a whole game), each playing whole plies until `budget_s` seconds have passed and it has played at least
    `min_plies` -- a bounded sample.  `value` weights the plies by GAME PHASE (seconds_per_game: mean ply time per 6-ply phase
    bin, summed over a 60.7-ply game) and carries a bootstrap 95 % interval over the streams (`value_ci95`); the plain
    plies-over-wall-clock figure of rounds 1-5 stays beside it as `value_unweighted`.  (One stream per hardware THREAD was
    measured and is slower: 256 streams on the box's 256 threads gave 0.32 games/s against 0.56-0.59 with 128 -- SMT siblings
    share the FMA units, and 256 private 12 MB weight sets thrash the last-level cache;
    profiles/r05_bench_driver_cmd_run1.json.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    try:
        hw_threads = len(os.sched_getaffinity(0))
    except AttributeError:
        hw_threads = os.cpu_count() or 1
    cores = hw_threads
    try:   # physical cores: hardware threads / SMT width of the machine
        import psutil
        logical, physical = psutil.cpu_count(logical=True), psutil.cpu_count(logical=False)
        if logical and physical and logical > physical:
            cores = max(1, hw_threads * physical // logical)
    except Exception:
        pass
    onet = ol.Net(net.num_blocks, net.num_filters, ol.state_dict_blob(net.state_dict()))
    cfg = ol.SelfplayCfg(sims, 15, 1, 1.0, 0.3, 0.25, 1, 0)
    plies = np.zeros(cores, dtype=np.int32)
    secs = np.zeros(cores, dtype=np.float64)
    cap = 256
    rec_phase = np.zeros((cores, cap), dtype=np.int32)
    rec_secs = np.zeros((cores, cap), dtype=np.float64)
    rec_evals = np.zeros((cores, cap), dtype=np.int32)
    ev, th, ended = C.c_int64(0), C.c_int(0), C.c_int64(0)
    i32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    t0 = time.time()
    n = ol.lib().orc_cpu_baseline_phased(onet.h, C.byref(cfg), cores, cores, int(min_plies), float(budget_s), 58, 42,
                                         plies.ctypes.data_as(i32p), secs.ctypes.data_as(f64p),
                                         C.byref(ev), C.byref(th), C.byref(ended), cap, rec_phase.ctypes.data_as(i32p),
                                         rec_secs.ctypes.data_as(f64p), rec_evals.ctypes.data_as(i32p))
    dt = time.time() - t0
    kept = np.minimum(plies, cap)
    value, (lo, hi) = phase_weighted_rate(rec_phase, rec_secs, kept, th.value)
    rate = plies / np.maximum(secs, 1e-9)            # plies/s of each stream on its own clock
    spg = seconds_per_game(np.concatenate([rec_phase[s, :kept[s]] for s in range(cores)]),
                           np.concatenate([rec_secs[s, :kept[s]] for s in range(cores)]))
    out = {
        "value": round(value, 5), "unit": "games/s", "cores": th.value, "streams": cores,
        "hardware_threads": hw_threads, "kind": "port",
        "value_ci95": [round(lo, 5), round(hi, 5)],
        "value_ci95_rel": round(max(value - lo, hi - value) / value, 4) if value > 0 else None,
        "estimator": "cores / seconds-per-game, seconds-per-game = sum over the %.1f plies of a game of the mean ply time of the "
                     "ply's %d-ply phase bin (%.1f s per game and stream here), so that cheap endgame plies are not over-"
                     "weighted; interval: bootstrap over the %d streams (2000 resamples, 2.5 / 97.5 percentiles)"
                     % (PLIES_PER_GAME, PHASE_BIN, spg, cores),
        "sample": "%d plies (%d network evals) of serial self-play, one stream per physical core (%d streams on %d OpenMP "
                  "threads), stream s starting 58*s/%d random plies into a game (phase-uniform: openings to endgames), %d-%d "
                  "plies each (>= %d, until %.0f s had passed), %dx%d net fp32, %d sims/move, %.1f s of wall clock"
                  % (n, ev.value, cores, th.value, cores, int(plies.min()), int(plies.max()), min_plies, budget_s,
                     net.num_blocks, net.num_filters, sims, dt),
        "evals_per_s": round(ev.value / dt, 1),
        "value_unweighted": round((n / PLIES_PER_GAME) / dt, 5),
        "per_stream_plies_per_s": {"min": round(float(rate.min()), 4), "median": round(float(np.median(rate)), 4),
                                   "max": round(float(rate.max()), 4),
                                   "note": "each stream's plies over its own busy time (late-starting streams play cheap endgame "
                                           "plies: not an error band of `value`)"},
    }
    if evals_per_game:
        out["value_by_evals"] = round(ev.value / dt / evals_per_game, 5)
    return out


def proportional_shares(rates, nominal, lanes):
    """Games per rank for the next step: proportional to the measured rates, every share within +-10 % of `nominal`
    and a multiple of `lanes`, job total exactly len(rates) * nominal.  Water-filling: ranks that hit a bound are fixed
    there and the rest of the total is re-divided among the others in proportion to their rates."""
    import numpy as np
    rate = np.asarray(rates, dtype=np.float64)
    world = len(rate)
    if not np.all(np.isfinite(rate)) or rate.min() <= 0:
        return [nominal] * world
    unit = lanes
    lo = int(np.ceil(0.9 * nominal / unit)) * unit
    hi = int(np.floor(1.1 * nominal / unit)) * unit
    total = nominal * world
    want = np.zeros(world)
    free = np.ones(world, dtype=bool)
    left = float(total)
    for _ in range(world + 1):
        if not free.any():
            break
        w = left * rate / rate[free].sum()
        over, under = free & (w > hi), free & (w < lo)
        if not over.any() and not under.any():
            want[free] = w[free]
            break
        fix = over if over.any() else under     # one side at a time: fixing both at once can strand part of the total
        want[fix] = hi if over.any() else lo
        left -= want[fix].sum()
        free &= ~fix
    new = [int(w) // unit * unit for w in want]
    # hand the rounding remainder out in units, fastest ranks first, never past the upper bound
    rem = total - sum(new)
    order = list(np.argsort(-rate))
    i = 0
    while rem >= unit and i < 4 * world:
        r = order[i % world]
        if new[r] + unit <= hi:
            new[r] += unit
            rem -= unit
        i += 1
    new[order[0]] += rem   # (only if every rank sits at the upper bound or nominal is not a multiple of the unit)
    return new


def union_ms(spans):
    """Total length of the union of (start, end) intervals (ms); `spans` = list of (n, 2) arrays (engine.union_ms)."""
    from othello_reinforcement_learning_test_amd.engine import union_ms as u
    return u(spans)


class Workload:
    """The engines of one configuration on this rank: `lanes` independent slot groups, each a SearchEngine on its own
    stream and host thread, sharing one set of packed weights; streaming self-play in steps (see the module docstring)."""

    def __init__(self, pkg, torch, board, blocks, filters, sims, games, lanes, stagger, step_games, rank=0,
                 precision=None, eval_cache=0, c_puct=1.0, temp_threshold=15):
        import threading
        self.pkg, self.torch, self.threading = pkg, torch, threading
        self.board, self.blocks, self.filters, self.sims = board, blocks, filters, sims
        self.games, self.lanes, self.step_games = games, max(1, lanes), step_games
        assert games % self.lanes == 0 and step_games % self.lanes == 0
        torch.manual_seed(42)
        self.net = pkg.OthelloResNet(blocks, filters, board_size=board).eval()
        self.ev = pkg.HipResNetEvaluator(self.net, precision=precision)
        self.engs = [pkg.SearchEngine(games // self.lanes, sims, temperature_threshold=temp_threshold, c_puct=c_puct,
                                      evaluator=self.ev, eval_cache_log2=eval_cache) for _ in range(self.lanes)]
        self.dev = torch.cuda.current_device()
        # the lanes' streams are made ONCE per process and device (engine.lane_streams: with new streams per workload the two
        # streams of every second two-lane workload shared a hardware queue and the workload lost 8-10 %;
        # OTHELLO_BENCH_NEW_STREAMS=1 restores that behaviour for the A/B, profiles/r05_lane_modes.log)
        if self.lanes > 1 and not os.environ.get("OTHELLO_BENCH_NEW_STREAMS"):
            self.streams = pkg.engine.lane_streams(self.lanes, self.dev)
        else:
            self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.lanes)] if self.lanes > 1 else [None]
        # lane overlap is a checked property: measured on the first warm-up step (play(check=True)); a
        # serialised arrangement is reported and the streams are drawn once more (engine.LaneOverlapCheck)
        self.check = pkg.engine.LaneOverlapCheck(self.lanes, self.dev,
                                                 max_redraws=int(os.environ.get("OTHELLO_LANE_REDRAWS", "1")))
        if os.environ.get("OTHELLO_NO_LANE_CHECK"):   # A/B hook: the warm-up runs without the HIP-event hooks (profiles/r06_lane_check_ab.log)
            self.check.pending = False
        # history ring per lane: room for the largest step target (+10 % rebalancing, + the games finishing while the
        # last rounds of a step are in flight) next to the games in flight
        per_lane = games // self.lanes
        hist = max(8 * per_lane, 2 * (int(1.1 * step_games) // self.lanes + 1) + 4 * per_lane)
        self.all_lanes(lambda e, k: e.stream_begin(42 + 1000003 * (rank * self.lanes + k), stagger_rounds=stagger,
                                                   hist_games=hist))

    def all_lanes(self, fn):
        """Run fn(engine, lane) for every lane -- each on its own stream and host thread -- and return the results."""
        torch, lanes = self.torch, self.lanes
        out, errors = [None] * lanes, []

        def work(k):
            try:
                torch.cuda.set_device(self.dev)   # the HIP current device is per thread and new threads start on device 0
                if self.streams[k] is None:
                    out[k] = fn(self.engs[k], k)
                else:
                    with torch.cuda.stream(self.streams[k]):
                        out[k] = fn(self.engs[k], k)
            except BaseException as exc:   # a failing lane must fail the whole bench, not leave stale tuples behind
                errors.append(exc)
        if lanes == 1:
            work(0)
        else:
            ths = [self.threading.Thread(target=work, args=(k,)) for k in range(lanes)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
        if errors:
            raise errors[0]
        return out

    def play(self, target, check=False):
        """One step's self-play: until >= target more games of this rank have finished -> (games, device tuple parts).
        check=True (warm-up steps only): if the lane-overlap check is still pending, this step runs with the HIP-event hooks
        on and decides it."""
        measuring = check and self.check.pending
        if measuring:
            self.check.begin(self.engs)
        try:
            return self._play(target)
        finally:
            if measuring and self.check.end(self.engs):
                self.torch.cuda.synchronize()      # the lanes move to other streams: nothing may be in flight on the old ones
                self.streams = self.pkg.engine.lane_streams(self.lanes, self.dev)

    def _play(self, target):
        # A launch of an fp16-split trunk that clamped an activation is never left standing (the reference's fp32 forward
        # has no clamp): every lane snapshots its stream before the step, and if the shared evaluator reports a clamp once
        # all lanes have joined it halves its activation scale and the step is replayed from the snapshots.  Seeded-random
        # weights (this benchmark) never saturate: the cost is a few MB of device copies per step and lane.
        while True:
            rescue = self.ev.precision != "f32"
            res = self.all_lanes(lambda e, k: ((e.snapshot() if rescue else None), e.stream_step(target // self.lanes))[1])
            if not (rescue and self.ev.needs_rescue()):
                break
            self.all_lanes(lambda e, k: e.restore())
        return sum(r[0] for r in res), [e_.selfplay_device_tensors() for e_ in self.engs]

    def counters(self):
        tot = {}
        for e_ in self.engs:
            for k, v in e_.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def cache_stats(self):
        tot = {}
        for e_ in self.engs:
            for k, v in e_.cache_stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def set_timing(self, on):
        for e_ in self.engs:
            e_.set_timing(on)

    def profiled_steps(self, n_steps, step_fn):
        """n_steps extra steps with the HIP-event hooks on -> the `prof` dict of the roofline objects."""
        prof = {"evals": 0, "net_ms": 0.0, "net_launches": 0, "tree_ms": 0.0, "tree_launches": 0, "union_ms": 0.0,
                "wall_s": 0.0, "games": 0}
        self.set_timing(True)
        for _ in range(n_steps):
            cp0 = self.counters()
            t1 = time.time()
            g = step_fn()
            prof["wall_s"] += time.time() - t1
            prof["games"] += g
            prof["evals"] += self.counters()["evals"] - cp0["evals"]
            spans = []
            for e_ in self.engs:
                for k, v in e_.kernel_time().items():
                    prof[k] += v
                spans.append(e_.net_spans())
            prof["union_ms"] += union_ms(spans)
        self.set_timing(False)
        return prof

    def roofline_numbers(self, prof):
        """-> (achieved TFLOP/s over the union of the trunk launches, peak, kernel info from the library)."""
        busy_s = prof["union_ms"] * 1e-3
        alg = prof["evals"] * mflop_per_position(self.blocks, self.filters, self.board) * 1e6
        achieved = alg / busy_s / 1e12 if busy_s > 0 else 0.0
        peak = PEAK_F32_TFLOPS if self.ev.precision == "f32" else PEAK_F16_TFLOPS
        info = self.ev.kernel_info(int(prof["evals"] / max(1, prof["net_launches"])))
        return achieved, peak, info

    def close(self):
        self.torch.cuda.synchronize()
        self.engs, self.ev, self.net = [], None, None
        import gc
        gc.collect()


def run_leg(pkg, torch, name, note="", board=8, blocks=10, filters=128, sims=50, games=4096, step_games=1536, warmup=2, steps=4,
            c_puct=1.0, temp_threshold=15, eval_cache=0, lanes=2, stagger=61):
    """A short secondary measurement in the same process (N = 1 only, after the headline's timed region and profiled
    step): the same streaming schedule on another BASELINE configuration -> a small dict for `other_configs`."""
    t_leg = time.time()
    w = Workload(pkg, torch, board, blocks, filters, sims, games, lanes, stagger, step_games, precision=None,
                 eval_cache=eval_cache, c_puct=c_puct, temp_threshold=temp_threshold)
    for _ in range(warmup):
        w.play(step_games, check=True)
    torch.cuda.synchronize()
    c0, t0, n = w.counters(), time.time(), 0
    cs0 = w.cache_stats() if eval_cache else None
    for _ in range(steps):
        n += w.play(step_games)[0]
    torch.cuda.synchronize()
    dt = time.time() - t0
    c1 = w.counters()
    st = {k: c1[k] - c0.get(k, 0) for k in c1}
    cs1 = w.cache_stats() if eval_cache else None
    prof = w.profiled_steps(1, lambda: w.play(step_games)[0])
    achieved, peak, info = w.roofline_numbers(prof)
    w.ev.check_saturation()
    out = {
        "config": name, "value": round(n / dt, 2), "unit": "games/s", "games_timed": n,
        "seconds_timed": round(dt, 2), "steps": steps, "warmup": warmup, "step_games": step_games,
        "concurrent_games": games, "lanes": lanes,
        "workload": "%dx%d, %d sims/move, %d-block x %d ResNet, c_puct %.2f, temperature threshold %d"
                    % (board, board, sims, blocks, filters, c_puct, temp_threshold),
        "evals_per_game": round(st["evals"] / max(1, st["games"]), 1),
        "plies_per_game": round(st["plies"] / max(1, st["games"]), 2),
        "kernel": info["kernel"], "dtype": w.ev.precision,
        "roofline_frac": round(achieved / peak, 4), "achieved_tflops": round(achieved, 2), "peak_tflops": peak,
        "frac_mfma_issue": round(achieved * info["issued_per_flop"] / peak, 4),
        "mfma_flops_issued_per_algorithmic_flop": round(info["issued_per_flop"], 3),
        "avg_launch_ms": round(prof["net_ms"] / max(1, prof["net_launches"]), 4),
        "positions_per_launch": round(prof["evals"] / max(1, prof["net_launches"]), 1),
        "net_time_share": round(prof["union_ms"] * 1e-3 / prof["wall_s"], 4) if prof["wall_s"] > 0 else None,
        "tree_kernels_ms": round(prof["tree_ms"], 2),
        # sum of the trunk launch durations / union of their intervals (profiled step): ~1 = the lanes' launches alternate
        # (two streams on one hardware queue), -> lanes when every lane always has a launch running
        "lanes_overlap": round(prof["net_ms"] / prof["union_ms"], 3) if prof["union_ms"] > 0 else None,
        "lanes_check_warmup": w.check.report(),
    }
    out["lanes_serialised"] = bool(lanes > 1 and out["lanes_overlap"] is not None and out["avg_launch_ms"] >=
                                   pkg.engine.OVERLAP_MIN_LAUNCH_MS and out["lanes_overlap"] < pkg.engine.overlap_floor(lanes))
    if eval_cache:
        hits = st.get("cache_hits", 0)
        cst = {k: cs1[k] - cs0[k] for k in ("distinct_positions", "repeated_evals", "conflict_evictions")}
        out["eval_cache"] = {"log2_entries_per_lane": eval_cache, "hits": hits,
                             "hit_rate": round(hits / max(1, hits + st["evals"]), 4),
                             "positions_reached_per_game": round((hits + st["evals"]) / max(1, st["games"]), 1),
                             # the misses, split: first evaluation of a position since the step's clear (compulsory) vs a
                             # position evaluated again (its entry was replaced in between, or a double miss in one launch)
                             "compulsory_misses": cst["distinct_positions"], "repeat_misses": cst["repeated_evals"],
                             "conflict_evictions": cst["conflict_evictions"],
                             "repeat_share_of_misses": round(cst["repeated_evals"] / max(1, st["evals"]), 4)}
    if note:
        out["note"] = note
    w.close()
    out["leg_seconds"] = round(time.time() - t_leg, 1)
    return out


def launch_ranks(n_ranks, argv):
    """--gpus N > 1 outside a torch.distributed.run environment: run the N ranks as a child process group and return its
    exit code.  This (parent) process never imports torch and never touches a GPU; it only relays -- the child inherits
    stdout (rank 0's JSON line) and stderr (heartbeats) -- and forwards SIGTERM / SIGINT to the whole group."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] --gpus %d without WORLD_SIZE: starting %d ranks as a child: %s" % (n_ranks, n_ranks, " ".join(cmd[1:9])),
          file=sys.stderr, flush=True)
    # the ranks inherit this process's environment as it is (HSA_ENABLE_IPC_MODE_LEGACY / GPU_MAX_HW_QUEUES: the caller's
    # values, or the defaults set at the top of this file) -- nothing is forced, an operator can flip either knob
    env = dict(os.environ, OTHELLO_BENCH_SELF_LAUNCHED="1")
    print("[bench] ranks' environment: HSA_ENABLE_IPC_MODE_LEGACY=%s GPU_MAX_HW_QUEUES=%s"
          % (env.get("HSA_ENABLE_IPC_MODE_LEGACY"), env.get("GPU_MAX_HW_QUEUES")), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)   # own group: torchrun and its ranks end together

    def forward(signum, _frame):
        try:
            os.killpg(proc.pid, signum)
        except ProcessLookupError:
            pass
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, forward)
    try:
        rc = proc.wait()
    finally:
        try:   # nothing of the group may outlive this process
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    return rc if rc >= 0 else 128 - rc


def first_contact_hint(stage, exc, rank=0, world=1):
    """What an operator should flip first when the N > 1 path fails at its first contact with RCCL: the text
    printed on stderr before the exception goes on.  No multi-GPU node was available to any round of this build, so the first
    N > 1 run is also the first time the IPC mode below is exercised."""
    return ("[bench rank %d/%d] %s failed: %r\n"
            "  This path has never run with more than one RCCL rank (no multi-GPU node was available to the build).  In order:\n"
            "  1. HSA_ENABLE_IPC_MODE_LEGACY is %s here (0 = dmabuf IPC handles, the only kind the build pool's host driver exports; the\n"
            "     symptom of the wrong mode is `hipIpcGetMemHandle: invalid argument`): try the other value, e.g.\n"
            "     HSA_ENABLE_IPC_MODE_LEGACY=%s python bench.py --gpus %d ...   (the self-launched ranks inherit it)\n"
            "  2. NCCL_DEBUG=INFO shows the transport RCCL chose and where it stopped.\n"
            "  3. OTHELLO_DIST_BACKEND=gloo runs the same control flow with the exchange on host copies (a line marked REHEARSAL): it\n"
            "     separates a transport problem from a problem of this code.\n"
            "  GPU_MAX_HW_QUEUES is %s (8 by default here; RCCL's own streams take hardware queues too: per_rank_lanes_overlap in the\n"
            "  JSON line shows whether the lanes still overlap)."
            % (rank, world, stage, exc, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
               "1" if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") == "0" else "0", world, os.environ.get("GPU_MAX_HW_QUEUES")))


def check_world(args, argv):
    """The launch environment against --gpus, BEFORE torch / the package / the GPU: returns None to go on as one rank of
    the job, or the exit code of the self-launched job."""
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and "RANK" not in os.environ:
        return launch_ranks(args.gpus, argv) if args.gpus > 1 else None
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but the launch environment has WORLD_SIZE=%s: start it as `python bench.py --gpus %d ...` "
              "(it launches its own ranks) or under `python -m torch.distributed.run --nproc-per-node %d`"
              % (args.gpus, env_world, args.gpus, args.gpus), file=sys.stderr, flush=True)
        raise SystemExit(2)
    return None


def summarize_legs(other_configs):
    """The secondary legs once more, compact: {configuration: [games/s, roofline frac, trunk share of the step, lanes overlap]} (or "skipped" /
    "error").  The driver keeps the last 2 000 characters of stdout, and the full `other_configs` entries are longer than
    that, so this goes LAST in the JSON line (< 300 characters)."""
    short = {"configs[1] + eval cache": "configs[1]+cache"}
    return {short.get(o["config"], o["config"]): ([o["value"], o["roofline_frac"], o["net_time_share"], o.get("lanes_overlap")] if "value" in o
                                                  else ("skipped" if "skipped" in o else "error"))
            for o in other_configs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--games", type=int, default=4096, help="concurrent game slots per GPU")
    ap.add_argument("--step-games", type=int, default=1536,
                    help="a step ends when this many more games per GPU have finished (steady state, slots stay full)")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--precision", default=None, help="f16x3 (default for 128 filters on 8x8), f16, f32 (exact, MFMA)")
    ap.add_argument("--board", type=int, default=8,
                    help="8 (the reference's game) or 6 (BASELINE configs[4], e.g. --board 6 --blocks 5 --filters 64 "
                         "--sims 25; rules parity UNPINNED: the reference has no 6x6 rules)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="independent game groups per GPU, each --games/--lanes slots on its own stream and host "
                         "thread (their kernels overlap: tails and tree phases of one lane are filled by the other)")
    ap.add_argument("--stagger", type=int, default=61,
                    help="slots join over this many ply rounds at stream start, so game phases are spread evenly and "
                         "games finish at a steady rate from the first timed step on (0: all at once)")
    ap.add_argument("--eval-cache", type=int, default=0,
                    help="log2 entries of the opt-in evaluation cache (0 = off; the headline number is measured with it OFF)")
    ap.add_argument("--c-puct", type=float, default=1.0)
    ap.add_argument("--temp-threshold", type=int, default=15)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short secondary legs (BASELINE configs[3], configs[4] and the evaluation-cache run of "
                         "configs[1]) that a default single-GPU run of the headline workload appends as `other_configs`")
    ap.add_argument("--equal-shares", action="store_true",
                    help="N>1: every rank targets exactly --step-games per step instead of rate-proportional shares")
    ap.add_argument("--profile-steps", type=int, default=1, help="extra steps after the timed region with HIP-event hooks on")
    ap.add_argument("--hooks-always", action="store_true",
                    help="keep the HIP-event hooks on for EVERY step (warm-up and timed region too) and report the mean "
                         "trunk launch duration over all launches of the run: the figure a rocprofv3 --kernel-trace "
                         "--stats of the same command must reproduce.  Not the headline configuration.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0,
                    help="seconds of CPU work for the baseline (every stream plays whole plies until this much time has passed "
                         "and it has played at least 8: ~35 s of wall clock at ~4 s per ply)")
    args = ap.parse_args()
    rc = check_world(args, sys.argv[1:])
    if rc is not None:
        sys.exit(rc)

    t_start = time.time()
    # (re)build the CPU oracle now: no fork+exec once this process holds the GPU.  Only the single-process run times the
    # CPU leg, so only it builds (eight ranks of an N>1 launch must not race one make target on a stale .so).
    if not args.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        oracle_lib.build()
    import numpy as np  # noqa: F401
    import torch

    import othello_reinforcement_learning_test_amd as pkg
    from othello_reinforcement_learning_test_amd import distributed as D

    try:
        rank, world, local = D.init_from_env()
    except Exception as exc:   # rendezvous / communicator creation: the first contact of an N > 1 job
        print(first_contact_hint("torch.distributed initialisation", exc, int(os.environ.get("RANK", "0")), args.gpus),
              file=sys.stderr, flush=True)
        raise
    assert world == args.gpus, (world, args.gpus)   # (check_world above)
    try:
        pkg._lib.require_device()   # no GPU => fail loudly
    except Exception as exc:
        raise SystemExit("bench.py rank %d/%d: %s [HSA_ENABLE_IPC_MODE_LEGACY=%s GPU_MAX_HW_QUEUES=%s]"
                         % (rank, world, exc, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), os.environ.get("GPU_MAX_HW_QUEUES")))
    import torch.distributed as dist
    if world > 1 and dist.get_backend() == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py rank %d/%d: %d ranks need %d GPUs, this node shows %d (RCCL needs one GPU per rank; "
                         "OTHELLO_DIST_BACKEND=gloo rehearses the control flow on fewer)"
                         % (rank, world, world, world, torch.cuda.device_count()))

    def beat(msg):
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.time() - t_start, msg), file=sys.stderr, flush=True)

    lanes = max(1, args.lanes)
    wl = Workload(pkg, torch, args.board, args.blocks, args.filters, args.sims, args.games, lanes, args.stagger,
                  args.step_games, rank=rank, precision=args.precision, eval_cache=args.eval_cache, c_puct=args.c_puct,
                  temp_threshold=args.temp_threshold)
    net, ev, engs, dev = wl.net, wl.ev, wl.engs, wl.dev

    # OTHELLO_FORCE_DIST=1 under torchrun --nproc-per-node 1: run the RCCL calls of the N>1 path on a one-rank group
    use_dist = world > 1 or (dist.is_available() and dist.is_initialized())
    gloo = use_dist and dist.get_backend() == "gloo"   # rehearsal mode: collectives on host copies

    def barrier():
        if use_dist:
            if gloo:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev])
        torch.cuda.synchronize()

    # Per-step target of each rank.  MI355X devices sustain clocks several per cent apart on this MFMA-dense work,
    # and the ranks meet at every step's all-gather, so equal targets would run the job at the pace of its slowest
    # GPU.  Each rank's target for the NEXT step follows its measured rate in the step before (computed identically
    # on every rank from one tiny all-gather, kept within +-10 % of equal, job total unchanged).
    nominal = args.step_games
    shares = [nominal] * world

    def rebalance(my_games, my_seconds):
        if args.equal_shares or not use_dist:
            return
        # (on a forced one-rank group the all-gather still runs -- it is the RCCL call an N>1 job makes -- and the
        # shares stay [nominal])
        t = torch.tensor([my_games / max(my_seconds, 1e-6)], dtype=torch.float64, device="cpu" if gloo else "cuda")
        allr = torch.zeros(world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allr, t)
        shares[:] = proportional_shares(allr.cpu().numpy(), nominal, lanes)

    xchg = {"s": 0.0, "n": 0}   # time of the exchange step inside the timed region (this rank)
    state = {"warming": True}   # warm-up steps may run the lane-overlap check (hooks on); timed steps never do

    def step():
        """-> (games this rank finished, replay samples of the whole job after the exchange)"""
        mine = shares[rank]
        t_play = time.time()
        games, parts = wl.play(mine, check=state["warming"])
        t_play = time.time() - t_play
        if use_dist:   # the one exchange step: RCCL all-gather of the replay tuples
            t_x = time.time()
            st, pi, z = ([p_[j].cpu() if gloo else p_[j] for p_ in parts] for j in range(3))   # one part per lane
            st, pi, z, _ = D.all_gather_replay(st, pi, z, force=True)   # persistent buffers: no per-step allocation
            samples = int(z.shape[0])
            torch.cuda.synchronize()   # (the collectives are asynchronous: the exchange is timed to its end)
            xchg["s"] += time.time() - t_x
            xchg["n"] += 1
        else:          # single GPU: the tuples stay where the lanes compacted them (no copy)
            samples = sum(int(p_[2].shape[0]) for p_ in parts)
        rebalance(games, t_play)
        return games, samples

    all_launch = {"ms": 0.0, "n": 0}
    if args.hooks_always:
        wl.set_timing(True)

    def tally_launches():
        if args.hooks_always:
            for e_ in engs:
                kt = e_.kernel_time()
                all_launch["ms"] += kt["net_ms"]
                all_launch["n"] += kt["net_launches"]

    counters = wl.counters
    beat("setup done (%dx%d net, %d slots in %d lanes, %d sims); warm-up: %d steps of %d games"
         % (args.blocks, args.filters, args.games, lanes, args.sims, args.warmup, args.step_games))
    try:
        barrier()   # also creates the RCCL communicator outside the timed region (matters when --warmup 0)
    except Exception as exc:
        if use_dist:
            print(first_contact_hint("the first barrier (communicator creation)", exc, rank, world), file=sys.stderr, flush=True)
        raise
    for i in range(args.warmup):
        t1 = time.time()
        try:
            g, _ = step()
        except Exception as exc:
            if use_dist and i == 0:   # the first exchange: counts all-gather + three padded all_gather_into_tensor
                print(first_contact_hint("the first step's exchange", exc, rank, world), file=sys.stderr, flush=True)
            raise
        tally_launches()
        beat("warm-up step %d/%d: %d games in %.2f s" % (i + 1, args.warmup, g, time.time() - t1))
    state["warming"] = False
    barrier()
    c0 = counters()
    xchg["s"], xchg["n"] = 0.0, 0
    t0 = time.time()
    my_games, samples = 0, 0
    for i in range(args.steps):
        t1 = time.time()
        g, samples = step()
        tally_launches()
        my_games += g
        beat("step %d/%d: %d games in %.2f s (%.1f games/s this rank, %.1f cumulative)"
             % (i + 1, args.steps, g, time.time() - t1, g / max(time.time() - t1, 1e-9), my_games / (time.time() - t0)))
    barrier()
    dt = time.time() - t0
    c1 = counters()
    total_games = my_games
    per_rank_rate, exchange_ms = [my_games / max(dt, 1e-9)], None
    if use_dist:
        t = torch.tensor([dt, float(my_games), xchg["s"] * 1e3 / max(1, xchg["n"])], dtype=torch.float64,
                         device="cpu" if gloo else "cuda")
        mine_t = t.clone()
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        allr = torch.zeros(3 * world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allr, mine_t)
        allr = allr.cpu().view(world, 3)
        per_rank_rate = [round(float(allr[r, 1] / max(float(allr[r, 0]), 1e-9)), 2) for r in range(world)]
        dt, total_games = float(tmax[0].item()), int(round(float(t[1].item())))
        exchange_ms = round(float(tmax[2].item()), 3)   # the slowest rank's mean exchange time per timed step
    stats = {k: c1[k] - c0.get(k, 0) for k in c1}

    # ---- profiled step(s) outside the timed region: HIP-event spans of every trunk launch ------------------------
    prof = {"evals": 0, "net_ms": 0.0, "net_launches": 0, "tree_ms": 0.0, "tree_launches": 0, "union_ms": 0.0,
            "wall_s": 0.0, "games": 0}
    if args.profile_steps > 0:
        def prof_step():
            g_ = step()[0]
            tally_launches()
            return g_
        prof = wl.profiled_steps(args.profile_steps, prof_step)
        if args.hooks_always:
            wl.set_timing(True)
        beat("profiled step: %d games in %.2f s, %d trunk launches, busy %.0f ms"
             % (prof["games"], prof["wall_s"], prof["net_launches"], prof["union_ms"]))
    ev.check_saturation()   # the clamp of the fp16-split trunk, surfaced: never a silent deviation from the reference
    # lane overlap of THIS rank in the profiled step (sum of the trunk launch durations / union of their intervals) and its
    # mean launch duration, gathered so that a slow rank of an N > 1 job is attributable: a longer launch = a lower clock,
    # a lower overlap = lanes sharing a hardware queue
    my_overlap = prof["net_ms"] / prof["union_ms"] if prof["union_ms"] > 0 else 0.0
    my_launch_ms = prof["net_ms"] / max(1, prof["net_launches"])
    per_rank_overlap, per_rank_launch_ms = [round(my_overlap, 3)], [round(my_launch_ms, 4)]
    if use_dist:
        t = torch.tensor([my_overlap, my_launch_ms], dtype=torch.float64, device="cpu" if gloo else "cuda")
        allo = torch.zeros(2 * world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allo, t)
        allo = allo.cpu().view(world, 2)
        per_rank_overlap = [round(float(allo[r, 0]), 3) for r in range(world)]
        per_rank_launch_ms = [round(float(allo[r, 1]), 4) for r in range(world)]

    if rank == 0:
        # With one lane the union equals the sum of the launch durations; with several lanes the launches of the
        # lanes overlap on the device, so FLOPs are divided by the time during which the kernel was running at all.
        net_s = prof["union_ms"] * 1e-3
        achieved, peak, kinfo = wl.roofline_numbers(prof)
        prec = ev.precision
        # HBM-side bytes per launch: the committed PMC passes over THIS command's own launch shape (two lanes, ~1 900
        # positions per trunk launch) -- a file constant, so it says when it no longer describes this build's kernel
        ct = committed_traffic(kinfo["kernel"])
        traffic, tree_traffic, traffic_basis = ct["traffic"], ct["tree_traffic"], ct["basis"]
        # which trunk kernel ran and how many MFMA FLOPs it issues per algorithmic FLOP come from the LIBRARY
        # (oth_net_kernel_info follows the dispatch of oth_net_forward_bits), not from a re-derivation here
        issued = kinfo["issued_per_flop"]
        wide = args.filters == 128 and args.board == 8 and prec != "f32"   # the configuration the PMC traffic file is for
        evals_per_game = stats["evals"] / max(1, stats["games"])
        out = {
            "metric": "self-play games/sec (%dx%d, %d MCTS sims/move)" % (args.board, args.board, args.sims),
            "value": round(total_games / dt, 3),
            "unit": "games/s",
            "n_gpus": world,
            "ranks": dist.get_world_size() if use_dist else 1,
            "backend": ("none (single process)" if not use_dist else
                        ("gloo (REHEARSAL on host copies)" if gloo else "nccl (= RCCL)")),
            "per_rank_games_per_s": per_rank_rate,
            "per_rank_lanes_overlap": per_rank_overlap,
            "per_rank_avg_launch_ms": per_rank_launch_ms,
            "exchange_ms_per_step": exchange_ms,
            # the runtime knobs the multi-lane / multi-rank paths depend on, as this rank saw them
            "runtime_env": pkg._lib.runtime_env(),
            "launched_by": ("bench.py itself (child torch.distributed.run)" if os.environ.get("OTHELLO_BENCH_SELF_LAUNCHED")
                            else ("torch.distributed.run of the caller" if "RANK" in os.environ else "single process")),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / max(1, args.steps) * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f16x3": "f16x3 (fp16 hi/lo operand split on MFMA, fp32 accumulate; fp32-equivalent)",
                      "f16": "f16 (single fp16 MFMA pass, fp32 accumulate)",
                      "f32": "f32 (exact fp32 on v_mfma_f32_16x16x4_f32)"}[prec],
            "data": "synthetic",
            "config": {
                "workload": "%dx%d, %d sims/move, %d-block x %d ResNet, %d concurrent games %s%s"
                            % (args.board, args.board, args.sims, args.blocks, args.filters, args.games,
                               "per rank, ranks SHARING one MI355X (gloo rehearsal)" if gloo else "on 1 MI355X per rank",
                               "" if args.board == 8 else " -- 6x6 RULES PARITY UNPINNED (the reference implements no 6x6 "
                               "game; checked against the 6x6 build of the CPU oracle only)"),
                "step": "steady-state streaming: a step ends when >= %d more games per GPU have finished (slots stay "
                        "full across steps; staggered start over %d ply rounds during warm-up)" % (args.step_games, args.stagger),
                "games_timed": total_games, "concurrent_games_per_gpu": args.games,
                "weights": "seeded random init (torch.manual_seed(42)), eval mode",
                "c_puct": args.c_puct, "temperature_threshold": args.temp_threshold, "dirichlet": "alpha 0.3 eps 0.25 (no effect on this search)",
                "parallelism": "dp%d: games sharded, %s" % (
                    world, "single GPU" if not use_dist else
                    ("gloo all-gather of replay tuples per step on host copies (REHEARSAL: several ranks share one GPU; "
                     "not a multi-GPU measurement)" if gloo else
                     "RCCL all-gather of replay tuples per step" + ("" if world > 1 else
                     " (one-rank group, OTHELLO_FORCE_DIST: exercises the RCCL calls of the N>1 path, not a multi-GPU "
                     "measurement)"))),
                "lanes_per_gpu": lanes,
                "step_targets_last_step": ("equal" if world == 1 or args.equal_shares else
                                           "proportional to each rank's measured rate in the previous step "
                                           "(+-10 %% of equal, job total fixed): %s" % list(shares)),
                "eval_cache": ("off (every position the search reaches is evaluated by the network)" if not args.eval_cache
                               else "ON: 2^%d entries, %d hits -- NOT the headline configuration" % (args.eval_cache, stats["cache_hits"])),
                "samples_last_step": samples,
                "evals_per_game": round(evals_per_game, 1),
                "plies_per_game": round(stats["plies"] / max(1, stats["games"]), 2),
                "timing_hooks_in_timed_region": bool(args.hooks_always),
            },
            "roofline": {
                "kernel": kinfo["kernel"],
                "bound": "mfma",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic if wide else None,
                "flops_basis": "ALGORITHMIC FLOPs of the direct 3x3 convolutions (%.2f MFLOP per position), whatever the "
                               "kernel issues: this kernel issues %.3f MFMA FLOPs per algorithmic FLOP (8x8, 128 filters: "
                               "the Winograd trunk 2.0, the direct one 2.75)"
                               % (mflop_per_position(args.blocks, args.filters, args.board), issued),
                "traffic_basis": traffic_basis,
                "traffic_stale": ct["stale"] if wide else None, "traffic_stale_why": ct["why"] if wide else None,
                "algorithmic_bytes_per_launch": round(prof["evals"] / max(1, prof["net_launches"]) * (24 + 4 * (args.board ** 2 + 2))),
                "measured_on": "%d profiled step(s) after the timed region (HIP-event hooks on, %d games, %.2f s)"
                               % (args.profile_steps, prof["games"], prof["wall_s"]),
                "frac_mfma_issue": round(achieved * issued / peak, 4),
                "launches": prof["net_launches"],
                "avg_launch_ms": round(prof["net_ms"] / max(1, prof["net_launches"]), 4),
                "avg_launch_ms_all_launches": (round(all_launch["ms"] / max(1, all_launch["n"]), 4)
                                               if args.hooks_always else None),
                "launches_all": all_launch["n"] if args.hooks_always else None,
                "busy_ms": round(prof["union_ms"], 1), "concurrent_lanes": lanes,
                # sum of the launch durations / union of their intervals: ~1 = the lanes' launches alternate (two streams on
                # one hardware queue: -8...-10 %), -> lanes when every lane always has a launch running
                "lanes_overlap": round(my_overlap, 3),
                "lanes_serialised": bool(lanes > 1 and my_launch_ms >= pkg.engine.OVERLAP_MIN_LAUNCH_MS
                                         and my_overlap < pkg.engine.overlap_floor(lanes)),
                "lanes_check_warmup": wl.check.report(),
                "time_basis": "union of the HIP-event intervals of all k_trunk launches (the %d lanes' launches "
                              "overlap; sum of launch durations = %.0f ms)" % (lanes, prof["net_ms"]),
                "positions_per_launch": round(prof["evals"] / max(1, prof["net_launches"]), 1),
                "mfma_flops_issued_per_algorithmic_flop": round(issued, 3),
                "net_time_share": round(net_s / prof["wall_s"], 4) if prof["wall_s"] > 0 else None,
                "tree_kernels_ms": round(prof["tree_ms"], 2), "tree_launches": prof["tree_launches"],
            },
            "cpu_baseline": None,
        }
        if wide and ct["stale"]:
            print("[bench] WARNING: roofline.traffic is stale (%s): re-run tools/bench_pmc.sh and commit its JSON under profiles/"
                  % ct["why"], file=sys.stderr, flush=True)
        # the rollout (tree search) kernel against the HBM roofline: algorithmic bytes of one launch (one simulation of
        # every game slot of a lane) / its mean duration, measured live in the profiled step (HIP events)
        tree_us = prof["tree_ms"] * 1e3 / max(1, prof["tree_launches"])
        tree_bytes = TREE_BYTES_PER_GAME_SIM * (args.games // lanes)
        tree_gbs = tree_bytes / (tree_us * 1e-6) / 1e9 if tree_us > 0 else 0.0
        out["roofline_rollout"] = {
            "kernel": "k_tree (expand + backup of the evaluated leaf, next PUCT descent or ply step; one launch per "
                      "simulation and lane)",
            "bound": "hbm", "achieved": round(tree_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(tree_gbs / PEAK_HBM_GBS, 5), "traffic": tree_traffic,
            "algorithmic_bytes_per_launch": tree_bytes,
            "bytes_basis": "%d B per game and simulation (24 B leaf out, 264 B result in, ~3 nodes x (32 B + 8 edges x 16 B) "
                           "read, ~3 path edges rewritten) x %d game slots per lane" % (TREE_BYTES_PER_GAME_SIM, args.games // lanes),
            "avg_launch_us": round(tree_us, 2), "launches": prof["tree_launches"],
            "note": "a dependent pointer chase through L2 / Infinity-Cache-resident trees (latency-bound, not bandwidth-"
                    "bound); it runs concurrently with the other lane's trunk launch",
        }
        # evidence first: if the CPU leg were to be killed the measured line is already on stderr
        print("[bench partial] " + json.dumps(out), file=sys.stderr, flush=True)
        # ---- short secondary legs: the other single-GPU BASELINE configurations, driver-run in the same process --------
        headline = (args.board, args.blocks, args.filters, args.sims, args.games, args.eval_cache, args.precision) == \
            (8, 10, 128, 50, 4096, 0, None)
        if headline and not use_dist and not args.no_other_configs:
            wl.close()
            engs = []
            # configs[3]: 4608 slots in THREE lanes of 1536.  The tree kernel of a 400-simulation search is ~0.4 ms per launch
            # beside a trunk launch of ~1.4 ms, so with two lanes there are moments when both are in it (trunk share of the
            # step 0.94-0.97, 76.7-77.3 games/s at 4096-4416 slots); the third lane keeps a trunk launch running (0.996;
            # profiles/r05_sweep_configs3.txt: 80.2 games/s, the same as 6624 slots in three lanes at a shorter ramp).
            # configs[4]: 8960 slots in FOUR lanes of 2240 -- k_trunk_w6 fills a CU with one workgroup, so the tree kernels of the
            # other lanes run in its shadow; a lane's launch is ~1 990 positions = 249 of 256 workgroups (2048 slots: 228), and
            # the fourth lane keeps a trunk launch running (trunk share 0.964 -> 0.978, +3.4 %: profiles/r05_sweep_configs4_b.txt).
            # configs[1] + cache: 8192 slots in two lanes and steps of 3072 games (the same slots : step ratio as the
            # headline, so a game sees as many cache clears), so that a launch of the MISSES is again ~1 350 positions
            # (profiles/r05_sweep_cache.txt: 1 663 vs 1 498 games/s at 4096 slots), and 2^24 entries per lane (4.9 GB of the
            # 288: repeated evaluations 9.8 % -> 5.0 % of the misses, +5 %: profiles/r05_sweep_cache_b.txt).
            legs = (
                dict(name="configs[3]", board=8, blocks=10, filters=128, sims=400, games=4608, lanes=3, step_games=255, warmup=2, steps=4,
                     c_puct=1.5, temp_threshold=20),
                dict(name="configs[4]", note="6x6 RULES PARITY UNPINNED (the reference implements no 6x6 game)",
                     board=6, blocks=5, filters=64, sims=25, games=8960, lanes=4, step_games=32768, warmup=3, steps=4),
                dict(name="configs[1] + eval cache", note="evaluation cache ON: NOT the headline configuration, never `value`",
                     board=8, blocks=10, filters=128, sims=50, games=8192, step_games=3072, warmup=3, steps=5, eval_cache=24),
            )
            out["other_configs"] = []
            for leg, (lname, lsec) in zip(legs, LEG_SECONDS):
                assert leg["name"] == lname
                rate_seen = max(out["value"], 1.0)
                need = 1.5 * lsec * MEASURED_RATE / rate_seen
                if time.time() - t_start + need > LEGS_HARD_STOP:
                    out["other_configs"].append({"config": lname, "skipped": "the run was %.0f s old and the leg is planned at %.0f s: "
                                                 "it would not end before %.0f s" % (time.time() - t_start, need / 1.5, LEGS_HARD_STOP)})
                    continue
                beat("leg %s ..." % lname)
                try:
                    out["other_configs"].append(run_leg(pkg, torch, **leg))
                    beat("leg %s: %.1f games/s, frac %.3f, %.0f s" % (lname, out["other_configs"][-1]["value"],
                                                                      out["other_configs"][-1]["roofline_frac"],
                                                                      out["other_configs"][-1]["leg_seconds"]))
                except Exception as exc:   # a failing leg must not take the headline down with it
                    out["other_configs"].append({"config": lname, "error": repr(exc)})
            print("[bench partial] " + json.dumps(out), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1 and args.board == 8:   # the oracle's CPU network is 8x8 only
            try:
                out["cpu_baseline"] = cpu_baseline(net, args.sims, args.cpu_budget, evals_per_game)
            except Exception as exc:   # the GPU result must still be reported
                out["cpu_baseline"] = {"error": repr(exc)}
        if out.get("other_configs"):
            out["other_configs_summary"] = summarize_legs(out["other_configs"])   # LAST key: survives the driver's tail
        print(json.dumps(out), flush=True)
    if use_dist:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
