"""bench.py's first contact with a multi-GPU node (BASELINE configs[2]; /root/reference/src/train/trainer.py:180-185 is the
single-process call it scales out) and the process hygiene of the GPU test session's launcher -- everything that can be
checked without a GPU: the launch environment is validated before torch or the package is imported, a plain
`bench.py --gpus N` starts its own N ranks as a child and relays their exit code, and the launcher's SIGTERM handler ends
every process under the stage it is running."""
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OTHELLO_FORCE_DIST")}
    env.update(kw)
    return env


def test_gpus_flag_must_match_the_launch_environment():
    """--gpus 2 with WORLD_SIZE=1 (or --gpus 1 under a 4-rank launch) is an error before anything heavy is imported --
    never a single-GPU run that prints "n_gpus": 1 for --gpus 2."""
    for gpus, world in (("2", "1"), ("8", "1"), ("1", "4")):
        t0 = time.time()
        r = subprocess.run([sys.executable, BENCH, "--gpus", gpus], env=_env(WORLD_SIZE=world, RANK="0"),
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and r.stdout == "" and "WORLD_SIZE=" + world in r.stderr, (gpus, world, r.stderr)
        assert time.time() - t0 < 20          # no torch import, no GPU call
    r = subprocess.run([sys.executable, BENCH, "--gpus", "0"], env=_env(), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and r.stdout == ""


def test_plain_gpus_n_launches_its_own_ranks_and_relays_their_exit_code():
    """No WORLD_SIZE + --gpus 2: bench.py starts torch.distributed.run --nproc-per-node 2 as a CHILD and exits with its
    code.  This container has no GPU, so both ranks fail loudly ("no gfx950 device") -- which is the point of the check:
    two ranks were started, each said which rank it was, the failure came back as a non-zero exit and no JSON line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(OTHELLO_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "starting 2 ranks as a child" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert "rank 0/2" in r.stderr and "rank 1/2" in r.stderr and "no gfx950" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_callers_ipc_mode_reaches_the_self_launched_ranks():
    """HSA_ENABLE_IPC_MODE_LEGACY (dmabuf vs legacy IPC handles: the knob RCCL's cross-process buffer registration depends on,
    never exercised with N > 1 ranks on this pool) is a DEFAULT in bench.py, not a force: an operator's own value -- and
    GPU_MAX_HW_QUEUES likewise -- must arrive in the ranks bench.py launches itself (line 463 used to
    overwrite it with "0").  Each rank prints the values it sees when it fails for want of a GPU."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=_env(OTHELLO_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="1", GPU_MAX_HW_QUEUES="6"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "ranks' environment: HSA_ENABLE_IPC_MODE_LEGACY=1 GPU_MAX_HW_QUEUES=6" in r.stderr       # the parent, before the launch
    for rank in (0, 1):                                                                              # ... and every rank
        line = [ln for ln in r.stderr.splitlines() if "rank %d/2" % rank in ln and "no gfx950" in ln]
        assert line and "[HSA_ENABLE_IPC_MODE_LEGACY=1 GPU_MAX_HW_QUEUES=6]" in line[0], r.stderr[-2000:]
    # unset by the caller: the defaults (dmabuf IPC, 8 hardware queues) arrive instead
    env = _env(OTHELLO_DIST_BACKEND="gloo")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    env.pop("GPU_MAX_HW_QUEUES", None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert "[HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8]" in r.stderr, r.stderr[-2000:]


def test_launcher_sigterm_ends_the_whole_stage_tree(tmp_path):
    """tests/gpu_children.py (the launcher of the GPU session's child stages) on SIGTERM: the running stage and everything
    under it -- here a grandchild in its OWN session, as bench.py's self-launched ranks are -- is gone, and the stage's rc
    file says 143 (killpg on the launcher's group used to miss the stages)."""
    pidfile = tmp_path / "grandchild.pid"
    stage = ("import subprocess, sys, time\n"
             "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'], start_new_session=True)\n"
             "open(%r, 'w').write(str(p.pid))\n"
             "time.sleep(600)\n" % str(pidfile))
    driver = ("import os, sys\n"
              "sys.path.insert(0, %r)\n"
              "import gpu_children as g\n"
              "g.main(%r, [('s', [sys.executable, '-c', %r], dict(os.environ))])\n"
              % (os.path.join(ROOT, "tests"), str(tmp_path / "out"), stage))
    launcher = subprocess.Popen([sys.executable, "-c", driver], start_new_session=True)
    try:
        t0 = time.time()
        while not pidfile.exists() or not pidfile.read_text():
            assert time.time() - t0 < 60 and launcher.poll() is None
            time.sleep(0.1)
        gpid = int(pidfile.read_text())
        os.kill(gpid, 0)                                   # alive
        launcher.send_signal(signal.SIGTERM)
        assert launcher.wait(timeout=60) == 143
        t0 = time.time()
        while True:
            try:
                os.kill(gpid, 0)
            except ProcessLookupError:
                break
            # (a zombie still answers kill 0 until it is reaped by init)
            state = open("/proc/%d/stat" % gpid).read().split(")")[-1].split()[0] if os.path.exists("/proc/%d/stat" % gpid) else "X"
            if state in ("Z", "X"):
                break
            assert time.time() - t0 < 30, "the stage's grandchild survived the launcher's SIGTERM"
            time.sleep(0.2)
        assert (tmp_path / "out" / "s.rc").read_text() == "143"
    finally:
        if launcher.poll() is None:
            launcher.kill()


def test_first_contact_hint_names_the_knobs(monkeypatch):
    """bench.py's N > 1 path has never met more than one RCCL rank; when its first contact fails (initialisation, the first
    barrier, the first exchange) the operator is told what to flip first -- the IPC mode (and to which value), NCCL_DEBUG, the gloo
    rehearsal -- instead of a bare traceback."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    t = b.first_contact_hint("the first barrier (communicator creation)", RuntimeError("hipIpcGetMemHandle: invalid argument"), 3, 8)
    assert "rank 3/8" in t and "the first barrier" in t and "hipIpcGetMemHandle" in t
    assert "HSA_ENABLE_IPC_MODE_LEGACY is 0 here" in t and "HSA_ENABLE_IPC_MODE_LEGACY=1 python bench.py --gpus 8" in t
    assert "NCCL_DEBUG=INFO" in t and "OTHELLO_DIST_BACKEND=gloo" in t and "GPU_MAX_HW_QUEUES is 8" in t
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    assert "HSA_ENABLE_IPC_MODE_LEGACY=0 python bench.py" in b.first_contact_hint("x", ValueError("y"), 0, 2)
    src = open(BENCH).read()
    assert src.count("first_contact_hint(") >= 4          # defined, and used at initialisation, the first barrier, the first exchange
