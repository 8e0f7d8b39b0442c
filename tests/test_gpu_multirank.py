"""BASELINE.json configs[2] (N-GPU data-parallel self-play with an all-gather of the replay tuples) kept from
rotting while no multi-GPU node is available: bench.py's N>1 path run by TWO ranks sharing GPU 0, collectives over
gloo (RCCL needs one GPU per rank).  The ranks are started by tests/conftest.py at session start, before this
process initialises the GPU; this test waits for them and checks the JSON line."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_rehearsal(rehearsal):
    p = rehearsal["proc"]
    assert p is not None, "the rehearsal was not started (run with `-m gpu`)"
    rc = p.wait(timeout=900)
    err = open(rehearsal["err"]).read()
    assert rc == 0, "rehearsal failed (rc %d):\n%s" % (rc, err[-3000:])
    lines = [ln for ln in open(rehearsal["out"]).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly one JSON line, got %d" % len(lines)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "games/s" and d["value"] > 0 and d["higher_is_better"] is True
    cfg = d["config"]
    assert cfg["games_timed"] >= 2 * 2 * 32           # both ranks' games are counted (whole-job aggregate)
    assert "dp2" in cfg["parallelism"] and "all-gather" in cfg["parallelism"]
    # the all-gathered tuple count of the last step covers both ranks: ~60 plies per game, >= 2 x 32 games
    assert cfg["samples_last_step"] > 2 * 32 * 40
    assert d["roofline"]["launches"] > 0 and d["cpu_baseline"] is None
    assert "step 2/2" in err                            # heartbeat lines on stderr
