#!/usr/bin/env python3
"""Static check of the shipped code objects for the hazards hipcc cannot see: the in-place MFMAs of the trunk kernels are
inline asm, so the compiler inserts no wait states between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB /
SrcC (2 wait states on gfx90a+; a v_mov zero-initialisation sunk next to the first MFMA of an accumulator chain gave 2e-3
errors in a 32-filter build of k_trunk_h3 before it was pinned), nor between an MFMA and an early consumer of its result.

Disassembles the gfx950 code object of every given .o (llvm-objdump) and fails if
  (1) a VALU instruction writes a source register of a v_mfma within the two preceding wait states, or
  (2) a non-MFMA instruction touches the result of an MFMA issued fewer than passes + 4 wait states earlier (8 for the
      4-pass 16x16x32 f16, 12 for the 8-pass shapes),
  (3) any packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) selects the HIGH dword
      of a source pair for its LOW lane (`op_sel:[..1..]`): on gfx950 that operand form returns wrong results now and then
      when two waves share a SIMD (round 4: tools/probes/probe_pk_opsel.hip, profiles/r04_heads_batch4_variants.txt -- the cause of the
      value-head errors of round 3).  hipcc's SLP vectoriser emits it for broadcast operands, so the library is built with
      -fno-slp-vectorize and its own packed arithmetic (net_epilogue.h) uses plain pairs only,
(1) and (2) along straight-line code AND ACROSS EVERY BRANCH EDGE: for each s_branch / s_cbranch_* the tail window of the branch's
block is carried into the head of the target block (so a VALU write at the end of a loop body followed by the back-edge to
an asm MFMA at the loop top is seen); fall-through edges are covered by the linear scan, which only resets at function
symbols.  A taken branch is counted as ONE wait state (conservative: it costs more).  Indirect jumps (s_setpc) are not
followed -- the trunk kernels have none (checked: the scan reports them).
usage: python tools/check_mfma_hazards.py [objects...]   (default: the trunk objects).  Called by __graft_entry__.build()."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# VGPRs v7 / v[4:7] and AGPRs a7 / a[4:7] (numbered 1000 + n: one unified file per lane, two name spaces): an MFMA may take
# A / B / C from either, and v_accvgpr_write is a VALU write like any other
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
WINDOW = 24
# XDL write of a VGPR -> VALU / LDS / VMEM read or overwrite of it: passes + 3 (+1 on gfx950) wait states, i.e. 12 for an
# 8-pass MFMA (cdna_hip_programming.md, inline-asm rules: "8-pass XDL: 12 states"), 8 for a 4-pass one.  Passes = pipe
# cycles / 4 (MI355X_MICROARCH.md cycle constants): 16x16x32 f16 / bf16 = 16 cycles = 4 passes -- hipcc's own hazard
# recogniser agrees: it places the first read of a builtin 16x16x32 result exactly 8 states behind it -- 32x32x16 and the
# fp32-input 16x16x4 = 32 cycles = 8 passes; anything else is taken as 16 passes.
def result_wait(mn):
    if re.search(r"_16x16x(32|16)_?(f16|bf16|bf8|fp8)", mn) or re.search(r"_16x16x(32|16)(f16|bf16)", mn):
        return 4 + 4
    if re.search(r"_32x32x16_?(f16|bf16)|_16x16x4_?f32|_32x32x8", mn):
        return 8 + 4
    return 16 + 4


def find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH / $HIP_PATH / /opt/rocm, then PATH."""
    cands = []
    for env in ("ROCM_PATH", "HIP_PATH"):
        if os.environ.get(env):
            cands.append(os.path.join(os.environ[env], "lib", "llvm", "bin", "llvm-objdump"))
            cands.append(os.path.join(os.environ[env], "llvm", "bin", "llvm-objdump"))
    cands += ["/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump"]
    for c in cands:
        if os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    w = shutil.which("llvm-objdump")
    if w:
        return w
    raise SystemExit("check_mfma_hazards: llvm-objdump not found (looked in $ROCM_PATH, $HIP_PATH, /opt/rocm and PATH): "
                     "the MFMA hazard check of the inline-asm trunk kernels cannot run")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(2)) + (1000 if m.group(1) == "a" else 0))
        else:
            base = 1000 if m.group(3) == "a" else 0
            out.update(range(base + int(m.group(4)), base + int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("addr", "mn", "ops", "opl", "ws", "dst", "func", "is_mfma", "target")

    def __init__(self, addr, mn, ops, func, target):
        self.addr, self.mn, self.ops, self.func, self.target = addr, mn, ops, func, target
        self.opl = [o.strip() for o in ops.split(",")] if ops else []
        self.is_mfma = mn.startswith("v_mfma") or mn.startswith("v_smfma")
        # wait states this instruction puts between its predecessors and its successors: s_nop N = N + 1; an MFMA = 4,
        # because the matrix pipe accepts one MFMA per 4 passes at best (16 cycles for v_mfma_f32_16x16x32_f16, measured:
        # MI355X_MICROARCH.md cycle constants; the 32x32 and fp32 shapes take 8), so whatever follows an intervening
        # MFMA issues at least 4 quad-cycles after the instruction in front of it; everything else = 1
        self.ws = int(self.opl[0], 0) + 1 if mn == "s_nop" else (4 if self.is_mfma else 1)
        self.dst = regs(self.opl[0]) if (mn.startswith("v_") and self.opl and not mn.startswith("v_cmp")) else set()

    def text(self):
        return "%s %s" % (self.mn, self.ops)


def disassemble(path):
    objdump = find_objdump()
    tmp = tempfile.mkdtemp(prefix="mfma_hz_")
    try:
        obj = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, obj)
        subprocess.check_call([objdump, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        cos = [f for f in glob.glob(obj + ".*") if "gfx950" in f]
        assert cos, "no gfx950 code object in " + path
        return subprocess.check_output([objdump, "-d", cos[0]], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def parse(text):
    """-> list of functions, each a list of Ins in address order."""
    funcs, cur, name = [], None, "?"
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            name, cur = m.group(1), []
            funcs.append(cur)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+): ([0-9A-Fa-f]{8})", ln)
        if not m or cur is None:
            continue
        mn, ops, addr, enc = m.group(1), m.group(2), int(m.group(3), 16), int(m.group(4), 16)
        target = None
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            simm = enc & 0xFFFF
            if simm >= 0x8000:
                simm -= 0x10000
            target = addr + 4 + 4 * simm
        cur.append(Ins(addr, mn, ops, name, target))
    return funcs


def hazards(seq, first_consumer, last_producer):
    """Scan `seq` (list of Ins); report hazards whose consumer index >= first_consumer and whose producer index <=
    last_producer (None: any)."""
    bad = []
    for i in range(first_consumer, len(seq)):
        c = seq[i]
        lo = max(0, i - WINDOW)
        if c.is_mfma:
            src = set()
            for o in c.opl[1:4]:
                src |= regs(o)
            dist = 0
            for j in range(i - 1, lo - 1, -1):
                if dist >= 2:
                    break
                p = seq[j]
                if (last_producer is None or j <= last_producer) and p.mn.startswith("v_") and not p.is_mfma and p.dst & src:
                    bad.append("%s: `%s` writes a source of `%s` %d wait state(s) earlier (0x%x -> 0x%x)"
                               % (c.func, p.text(), c.text(), dist, p.addr, c.addr))
                dist += p.ws
        elif c.opl and c.mn not in ("s_nop", "s_waitcnt", "s_barrier"):
            # the other direction: a non-MFMA instruction touching the result of a recent MFMA (XDL write -> VALU / LDS /
            # VMEM access of the same VGPR needs up to 11 wait states for a 4-pass op).  hipcc handles this for its own
            # MFMAs; flag it for any MFMA so that the asm ones are covered.
            touched = set()
            for o in c.opl:
                touched |= regs(o)
            dist = 0
            for j in range(i - 1, lo - 1, -1):
                if dist >= 20:
                    break
                p = seq[j]
                if (last_producer is None or j <= last_producer) and p.is_mfma and dist < result_wait(p.mn) and p.dst & touched:
                    bad.append("%s: `%s` touches the result of `%s` only %d wait state(s) later (0x%x -> 0x%x)"
                               % (c.func, c.text(), p.text(), dist, p.addr, c.addr))
                dist += p.ws
    return bad


# Rule (3).  Probed on the GPU beside an MFMA-issuing wave of the same SIMD (tools/probes/probe_pk_opsel.hip, round 4;
# tools/probes/probe_pk_opsel2.hip, round 5: profiles/r05_probe_pk_opsel.txt):
#   UNRELIABLE  v_pk_fma_f32 op_sel:[0,1,0], v_pk_mul_f32 op_sel:[0,1], v_pk_add_f32 op_sel:[0,1] -- SOURCE 1's high dword
#               routed to the low lane (37-47 % of wave-results wrong; 0 without an MFMA partner);
#   clean       v_pk_fma_f32 op_sel:[1,0,0] / [0,0,1] (source 0 / 2 high -> low), op_sel_hi:[0,1,1] / [1,0,1] (low -> high),
#               v_pk_mov_b32 op_sel:[1,0], v_pk_fma/mul/add_f16 with op_sel, v_fma_mixlo/mixhi_f16 with op_sel (the shipped
#               epilogue's form), packed fp32 with neg_lo / neg_hi on plain pairs (shipped).
# The rule stays wider than the defect: ANY high-to-low select on a packed-fp32 arithmetic instruction fails the build (the
# source-0 / source-2 routes were clean in one run of one instruction, v_pk_fma_f32 -- not enough to ship them unseen); the
# forms shown clean are counted per object so that what ships is known, not assumed (packed_forms below).
PK_F32 = re.compile(r"^v_pk_(fma|mul|add)_f32$")
PK_ANY = re.compile(r"^v_pk_\w+$|^v_fma_mix(lo|hi)?_f16$|^v_fma_mix_f32$")


def pk_opsel_hi_to_lo(fn):
    """Rule (3): packed-fp32 arithmetic whose op_sel routes a source's high dword to the low lane."""
    bad = []
    for x in fn:
        if PK_F32.match(x.mn):
            m = re.search(r"op_sel:\[([01,]+)\]", x.ops)
            if m and "1" in m.group(1):
                bad.append("%s: `%s` selects a source's HIGH dword for the low lane (unreliable on gfx950 with two waves "
                           "per SIMD) (0x%x)" % (x.func, x.text(), x.addr))
    return bad


def packed_forms(fn, counts):
    """Count every packed / mixed-precision VALU form of a function by mnemonic and modifier class."""
    for x in fn:
        if not PK_ANY.match(x.mn):
            continue
        m = re.search(r"op_sel:\[([01,]+)\]", x.ops)
        h = re.search(r"op_sel_hi:\[([01,]+)\]", x.ops)
        key = x.mn
        if m and "1" in m.group(1):
            key += " op_sel:[%s]" % m.group(1)
        if h and "0" in h.group(1) and not x.mn.startswith("v_fma_mix"):   # (mix ops: op_sel_hi = "this source is f16")
            key += " op_sel_hi:[%s]" % h.group(1)
        if "neg_lo" in x.ops or "neg_hi" in x.ops:
            key += " neg"
        counts[key] = counts.get(key, 0) + 1


def check_text(text, forms=None):
    """-> (number of MFMAs, hazard messages, number of branch edges checked, indirect jumps seen)"""
    n_mfma, bad, edges, indirect = 0, [], 0, 0
    for fn in parse(text):
        bad += pk_opsel_hi_to_lo(fn)
        if forms is not None:
            packed_forms(fn, forms)
        n_mfma += sum(1 for x in fn if x.is_mfma)
        indirect += sum(1 for x in fn if x.mn.startswith("s_setpc") or x.mn.startswith("s_swappc"))
        bad += hazards(fn, 0, None)                      # straight-line code incl. every fall-through edge
        index = {x.addr: k for k, x in enumerate(fn)}
        for b, x in enumerate(fn):
            if x.target is None or x.target not in index:
                continue
            t = index[x.target]
            if t == b + 1:
                continue                                 # same as the fall-through
            tail = fn[max(0, b - WINDOW + 1): b + 1]      # ... up to and including the branch
            head = fn[t: t + WINDOW]
            edges += 1
            bad += hazards(tail + head, len(tail), len(tail) - 1)
    seen, uniq = set(), []
    for m in bad:
        if m not in seen:
            seen.add(m)
            uniq.append(m)
    return n_mfma, uniq, edges, indirect


def check_object(path):
    n, bad, _, _ = check_text(disassemble(path))
    return n, bad


def main(paths):
    total, failed, forms = 0, [], {}
    for p in paths:
        n, bad, edges, indirect = check_text(disassemble(p), forms)
        total += n
        failed += bad
        print("%s: %d MFMA instructions, %d branch edges, %d hazards%s"
              % (os.path.relpath(p, ROOT), n, edges, len(bad),
                 "" if not indirect else " (%d indirect jumps NOT followed)" % indirect))
    per = {}
    for b in failed:
        per[b.split(":")[0]] = per.get(b.split(":")[0], 0) + 1
    for k, v in sorted(per.items()):
        print("  %4d in %s" % (v, k))
    for b in failed[:6]:
        print("  HAZARD " + b)
    # what ships: every packed / mixed-precision VALU form of the built objects (rule 3's comment says which were probed)
    sel = {k: v for k, v in forms.items() if "op_sel" in k or k.startswith("v_fma_mix")}
    plain = {k: v for k, v in forms.items() if k not in sel}
    print("packed forms in the built objects: " + (", ".join("%s x%d" % kv for kv in sorted(plain.items())) or "none"))
    print("  with an operand select: " + (", ".join("%s x%d" % kv for kv in sorted(sel.items())) or "none"))
    return 1 if failed else 0


if __name__ == "__main__":
    args = sys.argv[1:] or [os.path.join(ROOT, "othello_reinforcement_learning_test_amd", "csrc", f)
                            for f in ("net_mfma.o", "net_h3.o", "net_wino.o", "net_wino6.o", "net_f32.o", "engine.o",
                                      "replay_ops.o", "rules_api.o", "net.o")]
    sys.exit(main(args))
