"""Sweep of slot counts / lanes for one of bench.py's secondary legs (run_leg) in ONE GPU session, to size the leg:
usage: python tools/leg_sweep.py headline|configs3|configs4|cache  "games:lanes[:stagger[:step_games[:steps]]]" ...
prints one line per variant (games/s, roofline frac, positions per launch, trunk share of the step, tree ms, leg seconds)."""
import importlib.util
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import othello_reinforcement_learning_test_amd as pkg  # noqa: E402

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

BASE = {
    "headline": dict(name="configs[1]", board=8, blocks=10, filters=128, sims=50, warmup=4, steps=8, step_games=1536),
    "configs3": dict(name="configs[3]", board=8, blocks=10, filters=128, sims=400, c_puct=1.5, temp_threshold=20, warmup=2, steps=4,
                     step_games=256),
    "configs4": dict(name="configs[4]", board=6, blocks=5, filters=64, sims=25, warmup=3, steps=4, step_games=32769),
    "cache": dict(name="configs[1] + eval cache", board=8, blocks=10, filters=128, sims=50, warmup=5, steps=8, step_games=1536,
                  eval_cache=22),
}
which = sys.argv[1]
for var in sys.argv[2:]:
    f = var.split(":")
    kw = dict(BASE[which], games=int(f[0]), lanes=int(f[1]))
    if len(f) > 2 and f[2]:
        kw["stagger"] = int(f[2])
    if len(f) > 3 and f[3]:
        kw["step_games"] = int(f[3])
    if len(f) > 4 and f[4]:
        kw["steps"] = int(f[4])
    if len(f) > 5 and f[5]:
        kw["eval_cache"] = int(f[5])
    kw["step_games"] -= kw["step_games"] % kw["lanes"]
    r = bench.run_leg(pkg, torch, **kw)
    keep = ("value", "roofline_frac", "positions_per_launch", "net_time_share", "tree_kernels_ms", "avg_launch_ms", "leg_seconds",
            "evals_per_game", "games_timed", "seconds_timed", "eval_cache", "lanes_overlap", "lanes_serialised")
    print(var, json.dumps({k: r[k] for k in keep if k in r}), flush=True)
