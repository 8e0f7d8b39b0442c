"""A benchmarked workload itself, checked exactly -- at the SHAPE its number is quoted on (lanes, slots, stagger, step size).

bench.py's own Workload (the lanes of the leg as concurrent engines on their own streams and host threads, one set of packed
weights, streaming with the staggered start, bench.py's seeds) plays `--steps` steps of `--step-games` games; every game every
step returned is compared, tuple for tuple, with the CPU oracle's game of the same id (oracle driven by the HIP network's own
outputs), and each step's set of game ids with a host restatement of the streaming schedule computed from the oracle's game
lengths.  Reference: /root/reference/src/train/parallel_self_play.py:324-407 (the loop), configs/strong_8x8.yaml:29-31,
configs/debug_6x6.yaml:15-18 (the leg shapes' provenance).

usage (GPU box, repo root):
  python tools/bench_stream_exact.py                                  the headline: 8x8, 10x128, 50 sims, 4096 slots in 2 lanes
  python tools/bench_stream_exact.py --leg configs3                   bench.py's configs[3] leg: 400 sims, c_puct 1.5, thr 20,
                                                                      4608 slots in THREE lanes, one step of 255 games
  python tools/bench_stream_exact.py --leg configs4                   configs[4] leg: 6x6, 5x64, 25 sims, 8960 slots in FOUR lanes,
                                                                      one step of 32768 games -- against the 6x6 twin of the oracle
                                                                      (the reference has no 6x6 rules: PARITY UNPINNED)
  python tools/bench_stream_exact.py --leg cache                      configs[1] + evaluation cache (8192 slots, 2^24 entries)
  any of --board --blocks --filters --sims --c-puct --threshold --lanes --slots --stagger --step-games --steps --cache overrides.
Legacy form (rounds 2-5): `bench_stream_exact.py STEPS STEP_GAMES` with OTH_EXACT_CACHE / OTH_EXACT_SLOTS in the environment.

--oracle-games first (default): the oracle replays games 0 .. per-1 of every lane only when no refilled game finished inside the
checked steps (true for one step after a staggered start: a refilled game cannot end before the step does) -- asserted, never
assumed; otherwise (`all`) every game that was started (max id + 1 + slots per lane), as rounds 2-5 did.
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import othello_reinforcement_learning_test_amd as pkg     # noqa: E402
from test_gpu_selfplay_exact import simulate_stream      # noqa: E402  (the schedule restated on the host)

LEGS = {   # bench.py's `legs` tuple (bench.py: run_leg arguments), one checked step each
    "headline": dict(board=8, blocks=10, filters=128, sims=50, c_puct=1.0, threshold=15, lanes=2, slots=4096, stagger=61,
                     step_games=1536, steps=3, cache=0),
    "configs3": dict(board=8, blocks=10, filters=128, sims=400, c_puct=1.5, threshold=20, lanes=3, slots=4608, stagger=61,
                     step_games=255, steps=1, cache=0),
    "configs4": dict(board=6, blocks=5, filters=64, sims=25, c_puct=1.0, threshold=15, lanes=4, slots=8960, stagger=61,
                     step_games=32768, steps=1, cache=0),
    "cache": dict(board=8, blocks=10, filters=128, sims=50, c_puct=1.0, threshold=15, lanes=2, slots=8192, stagger=61,
                  step_games=3072, steps=2, cache=24),
}

ap = argparse.ArgumentParser()
ap.add_argument("legacy", nargs="*", help="[STEPS [STEP_GAMES]] (the rounds 2-5 form)")
ap.add_argument("--leg", default="headline", choices=sorted(LEGS))
for k, v in LEGS["headline"].items():
    ap.add_argument("--" + k.replace("_", "-"), type=type(v), default=None)
ap.add_argument("--oracle-games", default="first", choices=("first", "all"))
args = ap.parse_args()
cfg = dict(LEGS[args.leg])
for k in cfg:
    if getattr(args, k) is not None:
        cfg[k] = getattr(args, k)
if args.legacy:
    cfg["steps"] = int(args.legacy[0])
    if len(args.legacy) > 1:
        cfg["step_games"] = int(args.legacy[1])
if os.environ.get("OTH_EXACT_CACHE"):
    cfg["cache"] = int(os.environ["OTH_EXACT_CACHE"])
if os.environ.get("OTH_EXACT_SLOTS"):
    cfg["slots"] = int(os.environ["OTH_EXACT_SLOTS"])
board, lanes, slots, sims, thr, stagger = cfg["board"], cfg["lanes"], cfg["slots"], cfg["sims"], cfg["threshold"], cfg["stagger"]
steps, step_games, cache_log2 = cfg["steps"], cfg["step_games"], cfg["cache"]
per = slots // lanes
if board == 6:
    import oracle_lib6 as ol                               # noqa: E402  (PARITY UNPINNED: the 6x6 twin of the oracle)
else:
    import oracle_lib as ol                                # noqa: E402
U64 = np.uint64

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

print("leg %s: %dx%d, %dx%d net, %d sims/move, c_puct %.2f, threshold %d; %d slots in %d lanes of %d, stagger %d, %d step(s) of %d "
      "games; evaluation cache %s%s" % (args.leg, board, board, cfg["blocks"], cfg["filters"], sims, cfg["c_puct"], thr, slots, lanes,
                                         per, stagger, steps, step_games, "2^%d entries per lane" % cache_log2 if cache_log2 else "off",
                                         "" if board == 8 else "  [6x6: vs the 6x6 twin of the oracle, rules parity UNPINNED]"), flush=True)
t00 = time.time()
# bench.py's Workload: the network by torch.manual_seed(42), the lanes' engines / streams / host threads / seeds / history rings
w = bench.Workload(pkg, torch, board, cfg["blocks"], cfg["filters"], sims, slots, lanes, stagger, step_games, rank=0,
                   eval_cache=cache_log2, c_puct=cfg["c_puct"], temp_threshold=thr)
ev = w.ev
got = [[] for _ in range(lanes)]
t0 = time.time()
for i in range(steps):
    g, parts = w.play(step_games, check=(i == 0))
    torch.cuda.synchronize()
    for k, e in enumerate(w.engs):
        n = int(parts[k][2].shape[0])
        st, pi, z, gl = e.selfplay_fetch(n)
        got[k].append((e.game_ids(), st, pi, z, gl))
    print("step %d: %d games on the device, %.1f s since the stream began" % (i + 1, g, time.time() - t0), flush=True)
cnt = w.counters()
print("device: %d games, %d tuples, %d network evaluations, %d cache hits in %.1f s; lane check of step 1: %s"
      % (sum(len(x[0]) for lane in got for x in lane), sum(len(x[3]) for lane in got for x in lane), cnt["evals"],
         cnt["cache_hits"], time.time() - t0, w.check.report()), flush=True)


def dev_u64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=U64).view(np.int64)).cuda()


def fn(s, o):
    n = len(s)
    probs = np.empty((n, ol.NPOL), dtype=np.float32)
    vals = np.empty(n, dtype=np.float32)
    for i in range(0, n, per):
        m = min(per, n - i)
        ss, oo = np.zeros(per, dtype=U64), np.zeros(per, dtype=U64)
        ss[:m], oo[:m] = s[i:i + m], o[i:i + m]
        lg = ol.legal_batch(ss, oo)
        logp, v = ev.forward_bits(dev_u64(ss), dev_u64(oo), dev_u64(lg))
        probs[i:i + m] = ev.policy_probs(logp)[:m].cpu().numpy()
        vals[i:i + m] = v[:m, 0].cpu().numpy()
    return probs, vals


cb = ol.make_eval(fn)
ok_all, total_games, total_tuples = True, 0, 0
BIG = 10 ** 6    # "length" of a game the oracle did not replay: it can never finish inside the checked steps
for k in range(lanes):
    seed = 42 + 1000003 * (0 * lanes + k)                       # bench.py's stream seeds (Workload.__init__)
    max_id = max(int(x[0].max()) for x in got[k])
    first_only = args.oracle_games == "first" and max_id < per
    n_oracle = per if first_only else max_id + 1 + per              # lengths of the first generation / of every game started
    t1 = time.time()
    ws, wp, wz, wm, wl = ol.selfplay_philox(n_oracle, seed, sims, thr, cb, parallel_games=per, c_puct=cfg["c_puct"])
    print("lane %d: oracle replayed games 0..%d (%s) in %.1f s" %
          (k, n_oracle - 1, "the first generation: no refilled game finished in the checked steps" if first_only
           else "every game that was started", time.time() - t1), flush=True)
    if first_only:
        # the premise, checked on the oracle's own lengths: a refilled game starts when the first game of the lane ends and would
        # have to be over before the last checked step is -- i.e. be shorter than `window` plies; the shortest game replayed says
        # how far from possible that is (a refilled game this short would show up as an id >= per and fail the check below)
        fin = np.sort(np.array([(g * stagger) // per + int(wl[g]) for g in range(per)]))
        window = int(fin[min(len(fin), steps * (step_games // lanes)) - 1]) + 2 - int(fin[0])
        print("lane %d: refilled games had at most %d ply rounds before the last checked step ended; the shortest of the %d games "
              "replayed has %d plies" % (k, window, per, int(wl.min())), flush=True)
        if window >= int(wl.min()):
            print("lane %d: not a safe premise -> replaying every started game" % k, flush=True)
            first_only, n_oracle = False, max_id + 1 + per
            ws, wp, wz, wm, wl = ol.selfplay_philox(n_oracle, seed, sims, thr, cb, parallel_games=per, c_puct=cfg["c_puct"])
    woff = np.concatenate([[0], np.cumsum(wl)])
    lengths = np.concatenate([wl, np.full(4 * per + sum(len(x[0]) for x in got[k]), BIG, dtype=wl.dtype)])
    want_steps, _ = simulate_stream(lengths, per, stagger, [step_games // lanes] * steps)
    for i, ((ids, st, pi, z, gl), want_ids) in enumerate(zip(got[k], want_steps)):
        ok = ids.tolist() == want_ids and int(ids.max()) < n_oracle
        off = 0
        for gid, ln in zip(ids, gl):
            if gid >= n_oracle:
                ok = False
                break
            a, b = woff[gid], woff[gid + 1]
            ok &= bool(ln == wl[gid] and np.array_equal(st[off:off + ln], ws[a:b]) and
                       np.array_equal(pi[off:off + ln], wp[a:b]) and np.array_equal(z[off:off + ln], wz[a:b]))
            off += ln
        print("lane %d step %d: %d games (ids %d..%d), %d tuples: %s" %
              (k, i + 1, len(ids), ids.min(), ids.max(), len(z), "identical, ids as scheduled" if ok else "MISMATCH"), flush=True)
        ok_all &= ok
        total_games += len(ids)
        total_tuples += len(z)
w.close()
print("bench workload exact parity (leg %s%s): %s  (%d lanes x %d steps, %d games, %d tuples, %.0f s)"
      % (args.leg, "" if board == 8 else ", vs the 6x6 twin, rules unpinned", "IDENTICAL" if ok_all else "MISMATCH", lanes, steps,
         total_games, total_tuples, time.time() - t00))
sys.exit(0 if ok_all else 1)
