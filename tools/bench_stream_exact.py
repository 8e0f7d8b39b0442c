"""The benchmarked workload itself, checked exactly: bench.py's engines (two lanes of 2048 slots, 10x128 f16x3 trunk, 50
sims/move, threshold 15, streaming with the 61-round staggered start, bench.py's seeds) run `steps` steps of
`step_games` games; every game every step returned is compared, tuple for tuple, with the CPU oracle's game of the same
id (oracle driven by the HIP network's own outputs), and each step's set of game ids with a host restatement of the
streaming schedule computed from the oracle's game lengths.
usage (GPU box, repo root): python tools/bench_stream_exact.py [steps] [step_games]   -> profiles/rNN_bench_stream_exact.log
OTH_EXACT_CACHE=22 runs the engines with the opt-in evaluation cache (2^22 entries per lane): the tuples must STILL be identical
(a hit returns the bits an evaluation would have produced); the hit count is printed.  OTH_EXACT_SLOTS=8192 (with
OTH_EXACT_CACHE=24 and `2 3072` as arguments): the shape of bench.py's evaluation-cache leg since round 5."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                                   # noqa: E402
import othello_reinforcement_learning_test_amd as pkg     # noqa: E402
from test_gpu_selfplay_exact import simulate_stream      # noqa: E402  (the schedule restated on the host)

cache_log2 = int(os.environ.get("OTH_EXACT_CACHE", "0"))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
step_games = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
lanes, slots, sims, thr, stagger, rank = 2, int(os.environ.get("OTH_EXACT_SLOTS", "4096")), 50, 15, 61, 0
per = slots // lanes
U64 = np.uint64
torch.manual_seed(42)
net = pkg.OthelloResNet(10, 128).eval()
ev = pkg.HipResNetEvaluator(net)


def dev_u64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=U64).view(np.int64)).cuda()


def fn(s, o):
    n = len(s)
    probs = np.empty((n, 65), dtype=np.float32)
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
t00 = time.time()
for k in range(lanes):
    seed = 42 + 1000003 * (rank * lanes + k)                       # bench.py's stream seeds
    eng = pkg.SearchEngine(per, sims, temperature_threshold=thr, c_puct=1.0, evaluator=ev, eval_cache_log2=cache_log2)
    eng.stream_begin(seed, stagger_rounds=stagger, hist_games=8 * per)
    got, t0 = [], time.time()
    for _ in range(steps):
        g, n = eng.stream_step(step_games // lanes)
        st, pi, z, gl = eng.selfplay_fetch(n)
        got.append((eng.game_ids(), st, pi, z, gl))
    cnt = eng.counters()
    print("lane %d: %d steps, %d games, %d tuples on the device in %.1f s (eval cache %s: %d network evaluations, %d cache hits)" %
          (k, steps, sum(len(x[0]) for x in got), sum(len(x[3]) for x in got), time.time() - t0,
           "2^%d entries" % cache_log2 if cache_log2 else "off", cnt["evals"], cnt["cache_hits"]), flush=True)
    n_oracle = max(int(x[0].max()) for x in got) + 1 + per          # lengths of every game that was started
    t1 = time.time()
    ws, wp, wz, wm, wl = ol.selfplay_philox(n_oracle, seed, sims, thr, cb, parallel_games=per)
    print("lane %d: oracle replayed games 0..%d in %.1f s" % (k, n_oracle - 1, time.time() - t1), flush=True)
    woff = np.concatenate([[0], np.cumsum(wl)])
    want_steps, _ = simulate_stream(wl, per, stagger, [step_games // lanes] * steps)
    for i, ((ids, st, pi, z, gl), want_ids) in enumerate(zip(got, want_steps)):
        ok = ids.tolist() == want_ids
        off = 0
        for gid, ln in zip(ids, gl):
            a, b = woff[gid], woff[gid + 1]
            ok &= bool(ln == wl[gid] and np.array_equal(st[off:off + ln], ws[a:b]) and
                       np.array_equal(pi[off:off + ln], wp[a:b]) and np.array_equal(z[off:off + ln], wz[a:b]))
            off += ln
        print("lane %d step %d: %d games (ids %d..%d), %d tuples: %s" %
              (k, i + 1, len(ids), ids.min(), ids.max(), len(z), "identical, ids as scheduled" if ok else "MISMATCH"), flush=True)
        ok_all &= ok
        total_games += len(ids)
        total_tuples += len(z)
    del eng
print("bench workload exact parity: %s  (%d lanes x %d steps, %d games, %d tuples, %.0f s)"
      % ("IDENTICAL" if ok_all else "MISMATCH", lanes, steps, total_games, total_tuples, time.time() - t00))
sys.exit(0 if ok_all else 1)
