"""BASELINE.json configs[2] (N-GPU data-parallel self-play with an all-gather of the replay tuples) kept from
rotting while no multi-GPU node is available: bench.py's N>1 path run by FOUR ranks sharing GPU 0, collectives over
gloo (RCCL needs one GPU per rank; the box admits six processes on its card, so eight ranks cannot share it -- the
world-size-8 exchange step runs on CPU tensors in tests/test_distributed_cpu.py).  The ranks are started by tests/conftest.py at session start, before this
process initialises the GPU; this test waits for them and checks the JSON line."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_bench_multi_rank_rehearsal(rehearsal):
    p = rehearsal["proc"]
    assert p is not None, "the rehearsal was not started (run with `-m gpu`)"
    rc = p.wait(timeout=900)
    err = open(rehearsal["err"]).read()
    assert rc == 0, "rehearsal failed (rc %d):\n%s" % (rc, err[-3000:])
    lines = [ln for ln in open(rehearsal["out"]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line, got %d" % len(lines)
    d = json.loads(lines[0])
    w = rehearsal["ranks"]
    assert d["n_gpus"] == w and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "games/s" and d["value"] > 0 and d["higher_is_better"] is True
    cfg = d["config"]
    assert cfg["games_timed"] >= 2 * w * 29           # every rank's games are counted (shares within +-10 % of 32)
    assert "dp%d" % w in cfg["parallelism"] and "all-gather" in cfg["parallelism"]
    assert "REHEARSAL" in cfg["parallelism"] and "gloo" in cfg["parallelism"]    # never reads as an RCCL measurement
    # the all-gathered tuple count of the last step covers all ranks: ~60 plies per game, ~32 games per rank
    assert cfg["samples_last_step"] > w * 29 * 40
    assert d["roofline"]["launches"] > 0 and d["cpu_baseline"] is None
    rr = d["roofline_rollout"]           # the tree kernel against the HBM roofline, measured live in the profiled step
    assert rr["bound"] == "hbm" and rr["unit"] == "GB/s" and rr["achieved"] > 0 and rr["launches"] > 0
    assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-4
    assert "step 2/2" in err                            # heartbeat lines on stderr
