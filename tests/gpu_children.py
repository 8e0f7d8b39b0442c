#!/usr/bin/env python3
"""Launcher of the GPU test session's child processes (started by tests/conftest.py at session start, BEFORE the pytest
process touches the GPU: a GPU-initialised process must not fork+exec on this pool).  It never initialises the GPU itself
and runs its stages ONE AFTER THE OTHER, so that the card never holds more than 4 ranks + the pytest process (the box
admits six):

  stage "rehearsal"  bench.py --gpus 4 by FOUR gloo ranks sharing GPU 0 (the N>1 control flow);
  stage "rccl_bench" bench.py's N>1 path on a ONE-RANK nccl (= RCCL) group: OTHELLO_FORCE_DIST=1 under
                     torch.distributed.run --nproc-per-node 1 (RCCL needs one GPU per rank; the box has one);
  stage "rccl_worker" tests/rccl_one_rank_check.py on the same kind of group.

usage: gpu_children.py OUTDIR   -> OUTDIR/<stage>.out / .err / .rc, and OUTDIR/done when every stage has run."""
import os
import signal
import socket
import subprocess
import sys

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
        ("rehearsal", torchrun(REHEARSAL_RANKS, [bench, "--gpus", str(REHEARSAL_RANKS)] + SMALL), gloo),
        ("rccl_bench", torchrun(1, [bench, "--gpus", "1"] + SMALL), rccl),
        ("rccl_worker", torchrun(1, [os.path.join(ROOT, "tests", "rccl_one_rank_check.py")]), rccl),
    ]


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    for name, cmd, env in stages():
        with open(os.path.join(outdir, name + ".out"), "w") as out, open(os.path.join(outdir, name + ".err"), "w") as err:
            # own process group: on a timeout the whole stage (torchrun and its ranks) is ended, nothing by pattern
            proc = subprocess.Popen(cmd, stdout=out, stderr=err, env=env, cwd=ROOT, start_new_session=True)
            try:
                rc = proc.wait(timeout=900)
            except subprocess.TimeoutExpired:
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(proc.pid, sig)
                    except ProcessLookupError:
                        break
                    try:
                        proc.wait(timeout=20)
                        break
                    except subprocess.TimeoutExpired:
                        pass
                rc = 124
        with open(os.path.join(outdir, name + ".rc"), "w") as f:
            f.write(str(rc))
    open(os.path.join(outdir, "done"), "w").close()


if __name__ == "__main__":
    main(sys.argv[1])
