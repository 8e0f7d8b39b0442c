#!/usr/bin/env python3
"""Static check of the shipped code objects for the one hazard hipcc cannot see: the in-place MFMAs of net_mfma.hip /
net_h3.hip are inline asm, so the compiler inserts no wait states between a VALU write of a VGPR and an MFMA that reads
it as SrcA / SrcB / SrcC (2 wait states on gfx90a+; a v_mov zero-initialisation sunk next to the first MFMA of an
accumulator chain gave 2e-3 errors in a 32-filter build of k_trunk_h3 before it was pinned).  Disassembles the gfx950 code
object of every given .o (llvm-objdump) and fails if any VALU instruction writes a source register of a v_mfma within the
two preceding wait states.  usage: python tools/check_mfma_hazards.py [objects...]   (default: the two trunk objects)
Called by __graft_entry__.build()."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_object(path):
    tmp = tempfile.mkdtemp(prefix="mfma_hz_")
    try:
        obj = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, obj)
        subprocess.check_call([OBJDUMP, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        cos = [f for f in glob.glob(obj + ".*") if "gfx950" in f]
        assert cos, "no gfx950 code object in " + path
        text = subprocess.check_output([OBJDUMP, "-d", cos[0]], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    bad, n_mfma, func = [], 0, "?"
    window = []   # (wait states this instruction provides, mnemonic, dest regs, text)
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            func, window = m.group(1), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//", ln)
        if not m:
            continue
        mn, ops = m.group(1), m.group(2)
        opl = [o.strip() for o in ops.split(",")] if ops else []
        if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
            n_mfma += 1
            src = set()
            for o in opl[1:4]:
                src |= regs(o)
            dist = 0
            for ws, pmn, dst, ptxt in reversed(window):
                if dist >= 2:
                    break
                if pmn.startswith("v_") and not pmn.startswith("v_mfma") and dst & src:
                    bad.append("%s: `%s` writes a source of `%s %s` %d wait state(s) earlier" % (func, ptxt, mn, ops, dist))
                dist += ws
        elif opl and mn not in ("s_nop", "s_waitcnt", "s_barrier"):
            # the other direction: a non-MFMA instruction touching the result of a recent MFMA (XDL write -> VALU / LDS /
            # VMEM access of the same VGPR needs up to 11 wait states for a 4-pass op).  hipcc handles this for its own
            # MFMAs; flag it for any MFMA so that the asm ones are covered.
            touched = set()
            for o in opl:
                touched |= regs(o)
            dist = 0
            for ws, pmn, dst, ptxt in reversed(window):
                if dist >= 11:
                    break
                if pmn.startswith("v_mfma") and dst & touched:
                    bad.append("%s: `%s %s` touches the result of `%s` only %d wait state(s) later" % (func, mn, ops, ptxt, dist))
                dist += ws
        ws = 1
        if mn == "s_nop":
            ws = int(opl[0], 0) + 1
        dst = regs(opl[0]) if (mn.startswith("v_") and opl and not mn.startswith("v_cmp")) else set()
        window.append((ws, mn, dst, "%s %s" % (mn, ops)))
        window = window[-14:]
    return n_mfma, bad


def main(paths):
    total, failed = 0, []
    for p in paths:
        n, bad = check_object(p)
        total += n
        failed += bad
        print("%s: %d MFMA instructions, %d VALU->MFMA hazards" % (os.path.relpath(p, ROOT), n, len(bad)))
    per = {}
    for b in failed:
        per[b.split(":")[0]] = per.get(b.split(":")[0], 0) + 1
    for k, v in sorted(per.items()):
        print("  %4d in %s" % (v, k))
    for b in failed[:6]:
        print("  HAZARD " + b)
    return 1 if failed else 0


if __name__ == "__main__":
    args = sys.argv[1:] or [os.path.join(ROOT, "othello_reinforcement_learning_test_amd", "csrc", f)
                            for f in ("net_mfma.o", "net_h3.o", "net_wino.o", "net_wino6.o")]
    sys.exit(main(args))
