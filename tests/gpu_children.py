#!/usr/bin/env python3
"""Launcher of the GPU test session's child processes (started by tests/conftest.py at session start, BEFORE the pytest
process touches the GPU: a GPU-initialised process must not fork+exec on this pool).  It never initialises the GPU itself
and runs its stages ONE AFTER THE OTHER, so that the card never holds more than 4 ranks + the pytest process (the box
admits six):

  stage "rehearsal"  a PLAIN `python bench.py --gpus 4 ...` (OTHELLO_DIST_BACKEND=gloo): bench.py launches its own four
                     ranks as a child torch.distributed.run -- the first-contact path of an 8-GPU node -- and they share
                     GPU 0 (the N>1 control flow);
  stage "rccl_bench" bench.py's N>1 path on a ONE-RANK nccl (= RCCL) group: OTHELLO_FORCE_DIST=1 under
                     torch.distributed.run --nproc-per-node 1 (RCCL needs one GPU per rank; the box has one);
  stage "rccl_worker" tests/rccl_one_rank_check.py on the same kind of group.

usage: gpu_children.py OUTDIR   -> OUTDIR/<stage>.out / .err / .rc, and OUTDIR/done when every stage has run.

SIGTERM / SIGINT end the stage that is running -- every process under it, collected as exact PIDs from the process tree
BEFORE anything is signalled (a stage may hold further sessions: bench.py's self-launch starts its ranks in their own) --
write rc 143 and exit: an aborted pytest session never leaves ranks on the GPU."""
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REHEARSAL_RANKS = 4
SMALL = ["--steps", "2", "--warmup", "1", "--games", "64", "--step-games", "32", "--sims", "6", "--blocks", "2",
         "--filters", "16", "--stagger", "8", "--profile-steps", "1", "--no-cpu-baseline", "--no-other-configs"]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torchrun(nproc, script_args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
            "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + script_args


def stages():
    base = dict(os.environ)
    base.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "2"})
    base.pop("OTHELLO_FORCE_DIST", None)
    gloo = dict(base, OTHELLO_DIST_BACKEND="gloo")
    rccl = dict(base, OTHELLO_FORCE_DIST="1")
    rccl.pop("OTHELLO_DIST_BACKEND", None)
    bench = os.path.join(ROOT, "bench.py")
    return [
        ("rehearsal", [sys.executable, bench, "--gpus", str(REHEARSAL_RANKS)] + SMALL, gloo),   # self-launching
        ("rccl_bench", torchrun(1, [bench, "--gpus", "1"] + SMALL), rccl),
        ("rccl_worker", torchrun(1, [os.path.join(ROOT, "tests", "rccl_one_rank_check.py")]), rccl),
    ]


def descendants(pid):
    """Exact PIDs of every process under `pid` (children of children ..., whatever their session), pid itself first."""
    import psutil
    try:
        root = psutil.Process(pid)
        return [root] + root.children(recursive=True)
    except psutil.NoSuchProcess:
        return []


def end_tree(pid, grace=15.0):
    """SIGTERM to every process of the tree (collected first), SIGKILL to whatever is left after `grace` seconds."""
    import psutil
    procs = descendants(pid)
    for p in procs:
        try:
            p.terminate()
        except psutil.NoSuchProcess:
            pass
    _, alive = psutil.wait_procs(procs, timeout=grace)
    for p in alive:
        try:
            p.kill()
        except psutil.NoSuchProcess:
            pass
    psutil.wait_procs(alive, timeout=5.0)


CURRENT = {"proc": None, "name": None, "outdir": None}


def on_signal(signum, _frame):
    proc = CURRENT["proc"]
    if proc is not None and proc.poll() is None:
        end_tree(proc.pid)
    if CURRENT["name"]:
        with open(os.path.join(CURRENT["outdir"], CURRENT["name"] + ".rc"), "w") as f:
            f.write("143")
    os._exit(143)


def main(outdir, stage_list=None):
    os.makedirs(outdir, exist_ok=True)
    CURRENT["outdir"] = outdir
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, on_signal)
    for name, cmd, env in (stage_list if stage_list is not None else stages()):
        with open(os.path.join(outdir, name + ".out"), "w") as out, open(os.path.join(outdir, name + ".err"), "w") as err:
            proc = subprocess.Popen(cmd, stdout=out, stderr=err, env=env, cwd=ROOT, start_new_session=True)
            CURRENT["proc"], CURRENT["name"] = proc, name
            t0 = time.time()
            rc = None
            while rc is None:   # (a plain wait() would hold the signal handler off until the stage ends)
                try:
                    rc = proc.wait(timeout=1.0)
                except subprocess.TimeoutExpired:
                    if time.time() - t0 > 900:
                        end_tree(proc.pid)   # nothing by pattern: the exact PIDs of this stage's tree
                        rc = 124
            CURRENT["proc"] = None
        with open(os.path.join(outdir, name + ".rc"), "w") as f:
            f.write(str(rc))
        CURRENT["name"] = None
    open(os.path.join(outdir, "done"), "w").close()


if __name__ == "__main__":
    main(sys.argv[1])
