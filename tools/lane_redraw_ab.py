"""Same-box A/B of the lane-overlap check's REDRAW: does drawing the lanes' streams once more repair a
serialised arrangement, or is only the warning worth keeping?

The serialised arrangement is provoked the way round 5 found it (profiles/r05_lane_modes.log): with NEW streams per workload
(OTHELLO_BENCH_NEW_STREAMS=1) the two streams of every SECOND two-lane workload of a process land on one hardware queue.  Ten
identical two-lane legs (bench.py's run_leg, the headline's shape) with the redraw OFF (OTHELLO_LANE_REDRAWS=0: measure and
warn only), then ten with it ON (1): per leg the games/s, the overlap of the profiled step (sum of the trunk launch durations /
union of their intervals) and what the check saw on the warm-up steps.
usage (GPU box, repo root): python tools/lane_redraw_ab.py [legs_per_mode]   -> profiles/r06_lane_redraw_ab.log"""
import importlib.util
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OTHELLO_BENCH_NEW_STREAMS"] = "1"
import torch                                             # noqa: E402

import othello_reinforcement_learning_test_amd as pkg   # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
print("runtime_env:", json.dumps(pkg._lib.runtime_env()), flush=True)
for redraws in (0, 1):
    os.environ["OTHELLO_LANE_REDRAWS"] = str(redraws)
    print("== new streams per workload, OTHELLO_LANE_REDRAWS=%d (%s)" % (redraws, "measure and warn only" if redraws == 0 else
                                                                      "a serialised arrangement is drawn once more"), flush=True)
    for i in range(n):
        pkg.engine._WARNED.clear()
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            r = bench.run_leg(pkg, torch, "configs[1]", board=8, blocks=10, filters=128, sims=50, games=4096, lanes=2,
                              step_games=1536, warmup=2, steps=2)
        print("leg %d: %7.1f games/s  overlap %.2f  serialised %s  trunk share %.3f  launch %.3f ms  warm-up check %s  warnings %d"
              % (i + 1, r["value"], r["lanes_overlap"], r["lanes_serialised"], r["net_time_share"], r["avg_launch_ms"],
                 json.dumps(r["lanes_check_warmup"]), len([w for w in wl if "do not overlap" in str(w.message)])), flush=True)
