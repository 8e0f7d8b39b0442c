"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol the
header declares; the host-side single-board API (compiled from the same othello_rules.h as the
device kernels) matches the golden vectors and the oracle; device entry points fail loudly without a
GPU (no fallback).  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from othello_reinforcement_learning_test_amd import OthelloBitboard, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "othello_mi355x.h")).read()
    names = sorted(set(re.findall(r"\b(oth_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(_lib._SIGS), "ctypes table and header disagree"


def test_host_rules_vs_golden(golden):
    g = golden("g1_rules.npz")
    L = _lib.load()
    for arr, lcol, fcol, mcol in ((g["game_pos"], 2, 3, g["game_meta"][:, 0]),
                                  (g["crafted_pos"], 2, 3, g["crafted_meta"][:, 0])):
        for row, mv in zip(arr, mcol):
            s, o = int(row[0]), int(row[1])
            assert L.oth_legal_moves(s, o) == int(row[lcol])
            if mv < 64:
                assert L.oth_flip_bits(int(mv), s, o) == int(row[fcol])
    n, la, fa = (int(x) for x in g["checksum"])
    assert ol.rules_checksum(n) == (la, fa)


def test_board_object_follows_reference_games(golden):
    g = golden("g1_rules.npz")
    pos, meta, gid = g["game_pos"], g["game_meta"], g["game_id"]
    b = None
    for i in range(len(pos)):
        if i == 0 or gid[i] != gid[i - 1]:
            b = OthelloBitboard()
        assert (b.self_board, b.opp_board, b.move_count) == (int(pos[i, 0]), int(pos[i, 1]), int(meta[i, 3]))
        assert b.get_legal_moves_bits() == int(pos[i, 2])
        assert b.is_terminal() == bool(meta[i, 1])
        assert b.get_winner() == int(np.int8(meta[i, 2]))
        if not meta[i, 1]:
            lm = b.get_legal_moves()
            assert int(meta[i, 0]) in lm
            assert b.make_move(int(meta[i, 0])) is True
            assert b.passed == (meta[i, 0] == 64)


def test_board_invalid_moves_and_fields(golden):
    g = golden("g1_rules.npz")
    for (mv, ok, _s, _o, mc, passed), (s, o) in zip(g["invalid"], g["invalid_u64"]):
        b = OthelloBitboard()
        assert b.make_move(int(mv)) == bool(ok)
        assert (b.self_board, b.opp_board, b.move_count, int(b.passed)) == (int(s), int(o), mc, passed)
    b = OthelloBitboard()
    b.self_board, b.opp_board = 1 << 9, 1 << 8       # fields are writable (bitboard.pxd:25-28)
    assert (b.get_legal_moves_bits() >> 7) & 1        # H1 legal by wrap
    c = b.copy()
    c.make_move(7)
    assert b.self_board == 1 << 9 and c.move_count == 1  # copy independence (test_bitboard.py:179-192)
    assert b.get_stone_counts() == (1, 1)


def test_tensor_and_symmetries(golden):
    g = golden("g2_tensor.npz")
    rng = np.random.Generator(np.random.PCG64(3))
    for (s, o), t in zip(g["pos"], g["tensor"]):
        b = OthelloBitboard()
        b.self_board, b.opp_board = int(s), int(o)
        x = b.get_tensor_input()
        assert x.dtype == np.float32 and x.shape == (3, 8, 8) and x.flags["C_CONTIGUOUS"]
        assert np.array_equal(x, t.astype(np.float32))
    for (s, o) in g["pos"][:40]:
        b = OthelloBitboard()
        b.self_board, b.opp_board = int(s), int(o)
        pi = rng.random(65).astype(np.float32)
        st, ps = ol.symmetries(ol.board(s, o), pi)
        got = b.get_symmetries(pi)
        assert len(got) == 8
        for k in range(8):
            assert got[k][0].shape == (3, 8, 8) and got[k][1].shape == (65,)
            assert np.array_equal(got[k][0], st[k]) and np.array_equal(got[k][1], ps[k])


def test_repr_and_known_answers():
    b = OthelloBitboard()
    assert b.get_legal_moves() == [19, 26, 37, 44]            # reference tests/test_bitboard.py:29-37
    assert b.make_move(19) and b.get_stone_counts() == (1, 4)  # :60-71
    assert not b.make_move(19)                                 # :73-87
    r = b.to_string().split("\n")
    assert r[0] == "  A B C D E F G H" and len(r) == 9


def test_no_gpu_means_loud_failure():
    """Without an MI355X the compute classes must raise, never fall back to the CPU."""
    if _lib.device_available():
        pytest.skip("a GPU is present")
    from othello_reinforcement_learning_test_amd import (MCTS, OthelloHipError, OthelloResNet,
                                                         ParallelSelfPlayWorker, SearchEngine)
    with pytest.raises(OthelloHipError):
        SearchEngine(4, 5)
    net = OthelloResNet(2, 16)
    with pytest.raises(OthelloHipError):
        MCTS(net)
    with pytest.raises(OthelloHipError):
        ParallelSelfPlayWorker(OthelloBitboard, net, num_parallel_games=2, num_simulations=2)
    out = np.zeros(4, dtype=np.uint64)
    assert _lib.load().oth_legal_moves_batch(out.ctypes.data, out.ctypes.data, out.ctypes.data, 4, None) == -1
    assert "no gfx950" in _lib.last_error()
    # the round-5 entry points too: a network handle can be made on the host, but nothing that would touch its device
    # arrays -- the activation scale, the engine snapshot, the cache statistics -- runs without a device
    import ctypes as C
    L = _lib.load()
    h = L.oth_net_create(2, 16, 8)
    assert h
    s = C.c_float(0)
    assert L.oth_net_get_act_scale(h, C.byref(s)) == 0 and s.value == 16.0           # (a plain host field)
    assert L.oth_net_set_act_scale(h, C.c_float(8.0)) != 0 and "no gfx950" in _lib.last_error()
    L.oth_net_destroy(h)
    stats = (C.c_int64 * 4)()
    for rc in (L.oth_engine_snapshot(None, None), L.oth_engine_restore(None, None), L.oth_engine_cache_stats(None, stats, None)):
        assert rc != 0


def test_header_is_plain_c_and_links(tmp_path):
    """include/othello_mi355x.h must be consumable by a C compiler (the drop-in boundary is a C ABI)
    and every declared entry point must resolve against the shared library: a C program that takes the address of
    each one is compiled with gcc -std=c99 -pedantic, linked against libothello_mi355x.so and run.  It also plays a few
    moves through the host board API (no device needed) and checks the no-device error path of a device call."""
    import re
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "othello_mi355x.h")).read()
    names = sorted(set(re.findall(r"\b(oth_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40
    src = tmp_path / "abi.c"
    table = ",\n".join('    {"%s", (void (*)(void))%s}' % (n, n) for n in names)
    src.write_text('''
#include <stdio.h>
#include <string.h>
#include "othello_mi355x.h"
struct entry { const char *name; void (*fn)(void); };
static const struct entry table[] = {
%s
};
int main(void) {
    size_t i, n = sizeof(table) / sizeof(table[0]);
    oth_board b;
    uint64_t legal;
    for (i = 0; i < n; ++i) if (!table[i].fn) { printf("null %%s\\n", table[i].name); return 2; }
    oth_board_reset(&b);
    legal = oth_legal_moves(b.self_board, b.opp_board);
    if (legal != 0x0000102004080000ULL) { printf("legal %%llx\\n", (unsigned long long)legal); return 3; }
    if (!oth_board_make_move(&b, 19) || oth_board_make_move(&b, 19) || b.move_count != 1) return 4;
    if (oth_board_is_terminal(&b)) return 5;
    if (!oth_device_available()) {   /* CPU-only host: device calls must fail loudly, never fall back */
        uint64_t s = b.self_board, o = b.opp_board, out = 0;
        if (oth_legal_moves_batch(&s, &o, &out, 1, NULL) == 0) return 6;
        if (strlen(oth_last_error()) == 0) return 7;
    }
    printf("ok %%d entry points, %%s\\n", (int)n, oth_version());
    return 0;
}
''' % table)
    libdir = os.path.join(root, "othello_reinforcement_learning_test_amd")
    exe = tmp_path / "abi"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src),
           "-o", str(exe), "-L", libdir, "-l:libothello_mi355x.so", "-Wl,-rpath," + libdir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("ok %d entry points" % len(names))


def test_integration_md_cfg_struct_matches_the_header():
    """The ctypes struct printed in INTEGRATION.md must have exactly the fields of oth_engine_cfg (a shorter struct
    would make oth_engine_create read past the caller's memory)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"class Cfg\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n", md, flags=re.S)
    doc_fields = re.findall(r'\("(\w+)", C\.(\w+)\)', m.group(1))
    from othello_reinforcement_learning_test_amd import _lib
    import ctypes as C
    lib_fields = [(n, t) for n, t in _lib.EngineCfg._fields_]
    assert [(n, getattr(C, t)) for n, t in doc_fields] == lib_fields
    hdr = open(os.path.join(root, "include", "othello_mi355x.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} oth_engine_cfg;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hdr_fields = re.findall(r"(?:int32_t|float|double)\s+(\w+);", body)
    assert hdr_fields == [n for n, _ in lib_fields]


def test_no_mfma_hazards_in_built_objects():
    """The in-place MFMAs of the trunk kernels are inline asm, invisible to hipcc's hazard recogniser: the built gfx950
    code objects must have no VALU write of an MFMA source within two wait states (and no early consumer of an MFMA
    result).  tools/check_mfma_hazards.py disassembles net_mfma.o / net_h3.o; it needs llvm-objdump from /opt/rocm."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = [os.path.join(root, "othello_reinforcement_learning_test_amd", "csrc", f)
            for f in ("net_mfma.o", "net_h3.o", "net_wino.o", "net_wino6.o", "net_f32.o")]
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(root, "tools", "check_mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("object files not present (the .o files do not travel to the GPU box)")
    try:
        mod.find_objdump()          # $ROCM_PATH / $HIP_PATH / /opt/rocm / PATH (not one hard-coded place)
    except SystemExit:
        pytest.skip("llvm-objdump not found")
    forms = {}
    for o in objs:
        n, bad, _, _ = mod.check_text(mod.disassemble(o), forms)
        assert n > 200 and not bad, bad[:3]
    # rule 3's inventory: the only operand-select forms that ship are the epilogue's v_fma_mixlo / mixhi_f16 (probed clean
    # beside an MFMA partner, profiles/r05_probe_pk_opsel.txt); packed fp32 ships on plain pairs only
    sel = sorted(k for k in forms if "op_sel" in k)
    assert sel == ["v_fma_mixhi_f16 op_sel:[1,0,0]"], sel
    assert forms.get("v_fma_mixlo_f16", 0) == forms["v_fma_mixhi_f16 op_sel:[1,0,0]"] > 0


def test_hazard_checker_sees_across_branch_edges():
    """The checker on synthetic disassembly (round-3 advisor finding: it only modelled straight-line code): a VALU write
    at the END of a loop body followed by the back-edge to an asm MFMA at the loop top, an MFMA result consumed right
    after a forward branch, the same two patterns made safe by wait states, and intervening MFMAs counted as 4 states."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", os.path.join(root, "tools", "check_mfma_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def asm(lines):
        """[(mnemonic, operands)] -> objdump-style text with addresses; ('L', name) places a label, branches name one."""
        addr, labels, rows = 0x1000, {}, []
        for mn, ops in lines:
            if mn == "L":
                labels[ops] = addr
                continue
            rows.append((addr, mn, ops))
            addr += 8 if mn.startswith("v_mfma") else 4
        out = ["0000000000001000 <k>:"]
        for a, mn, ops in rows:
            enc = 0xBF800000
            if mn.startswith("s_cbranch") or mn == "s_branch":
                enc = 0xBF850000 | (((labels[ops] - a - 4) // 4) & 0xFFFF)
                ops = str((labels[ops] - a - 4) // 4)
            out.append("\t%s %s // %012X: %08X" % (mn, ops, a, enc))
        return "\n".join(out)
    mfma = ("v_mfma_f32_16x16x32_f16", "v[0:3], v[8:11], v[12:15], v[0:3]")
    # (1) loop: top = asm MFMA reading v8; bottom = VALU write of v8, then the back-edge
    loop_bad = asm([("L", "top"), mfma, ("s_nop", "7"), ("s_nop", "7"), ("v_add_f32_e32", "v8, v20, v21"),
                    ("s_cbranch_scc1", "top"), ("s_endpgm", "")])
    n, bad, edges, _ = mod.check_text(loop_bad)
    assert n == 1 and edges == 1 and len(bad) == 1 and "writes a source" in bad[0]
    loop_ok = asm([("L", "top"), mfma, ("s_nop", "7"), ("s_nop", "7"), ("v_add_f32_e32", "v8, v20, v21"), ("s_nop", "0"),
                   ("s_cbranch_scc1", "top"), ("s_endpgm", "")])
    assert mod.check_text(loop_ok)[1] == []
    # (2) forward branch: the MFMA's result read 2 states later at the target; linear scan alone would not see it
    fwd_bad = asm([mfma, ("s_branch", "out"), ("s_nop", "7"), ("s_nop", "7"), ("L", "out"), ("v_mov_b32_e32", "v30, v1"),
                   ("s_endpgm", "")])
    n, bad, edges, _ = mod.check_text(fwd_bad)
    assert edges == 1 and len(bad) == 1 and "touches the result" in bad[0]
    fwd_ok = asm([mfma, ("s_nop", "6"), ("s_branch", "out"), ("s_nop", "0"), ("L", "out"),
                  ("v_mov_b32_e32", "v30, v1"), ("s_endpgm", "")])
    assert mod.check_text(fwd_ok)[1] == []
    # (3) two other MFMAs in between = 8 states (the pipe takes one MFMA per 4 passes): enough for the 4-pass 16x16x32,
    #     not for an 8-pass 32x32x16 result (12)
    other = ("v_mfma_f32_16x16x32_f16", "v[4:7], v[8:11], v[12:15], v[4:7]")
    assert mod.check_text(asm([mfma, other, other, ("v_mov_b32_e32", "v30, v1")]))[1] == []
    assert len(mod.check_text(asm([mfma, other, ("v_mov_b32_e32", "v30, v1")]))[1]) == 1
    big = ("v_mfma_f32_32x32x16_f16", "v[0:15], v[16:19], v[20:23], v[0:15]")
    assert len(mod.check_text(asm([big, other, other, ("v_mov_b32_e32", "v30, v1")]))[1]) == 1
    assert mod.check_text(asm([big, other, other, other, ("v_mov_b32_e32", "v30, v1")]))[1] == []
    assert mod.find_objdump()
    # (4) rule 3: a packed-fp32 op taking a source's HIGH dword for its low lane (the round-3 value-head defect)
    pk = lambda sel: asm([("v_pk_fma_f32", "v[4:5], v[22:23], v[14:15], v[4:5] " + sel)])   # noqa: E731
    assert len(mod.check_text(pk("op_sel:[0,1,0]"))[1]) == 1
    assert mod.check_text(pk("op_sel_hi:[1,0,1]"))[1] == [] and mod.check_text(pk(""))[1] == []
    assert mod.check_text(asm([("v_pk_fma_f16", "v4, v22, v14, v4 op_sel:[0,1,0]")]))[1] == []   # 16-bit packed ops: probed clean (r5)
    # round 5's probe: v_pk_mul / v_pk_add with source 1's high dword fail like the FMA; the rule also keeps the source-0 /
    # source-2 routes out (clean in one run -- not shipped unseen); v_pk_mov_b32 was probed clean and is only counted
    for mn, ops in (("v_pk_mul_f32", "v[4:5], v[22:23], v[14:15] op_sel:[0,1]"), ("v_pk_add_f32", "v[4:5], v[22:23], v[14:15] op_sel:[0,1]"),
                    ("v_pk_fma_f32", "v[4:5], v[22:23], v[14:15], v[4:5] op_sel:[1,0,0]")):
        assert len(mod.check_text(asm([(mn, ops)]))[1]) == 1, mn
    forms = {}
    assert mod.check_text(asm([("v_pk_mov_b32", "v[4:5], v[22:23], v[14:15] op_sel:[1,0]"),
                               ("v_pk_add_f32", "v[4:5], v[22:23], v[14:15] neg_lo:[0,1] neg_hi:[0,1]")]), forms)[1] == []
    assert forms == {"v_pk_mov_b32 op_sel:[1,0]": 1, "v_pk_add_f32 neg": 1}
    # (5) AGPR operands: v_accvgpr_write of an MFMA's A operand right in front of it; an AGPR result read too early
    amf = ("v_mfma_f32_16x16x32_f16", "v[0:3], a[16:19], v[12:15], v[0:3]")
    assert len(mod.check_text(asm([("v_accvgpr_write_b32", "a17, v40"), amf]))[1]) == 1
    assert mod.check_text(asm([("v_accvgpr_write_b32", "a17, v40"), ("s_nop", "1"), amf]))[1] == []
    assert mod.check_text(asm([("v_accvgpr_write_b32", "a21, v40"), amf]))[1] == []           # another register
    agm = ("v_mfma_f32_16x16x32_f16", "a[0:3], v[8:11], v[12:15], a[0:3]")
    assert len(mod.check_text(asm([agm, ("v_accvgpr_read_b32", "v9, a2")]))[1]) == 1


def test_zero_simulation_policy_and_force_dist_parsing(monkeypatch):
    """Two small host-side behaviours.  (1) node.py:162-182 with an expanded but unvisited root: at T = 0 the
    reference returns the one-hot on actions[argmax(zeros)] = the FIRST child; at T != 0 it evaluates 0/0, where this mirror
    returns zeros (documented deviation, parity unpinned).  (2) OTHELLO_FORCE_DIST=0 / false means OFF."""
    import numpy as np
    from othello_reinforcement_learning_test_amd import distributed as D
    from othello_reinforcement_learning_test_amd.engine import policy_from_visits
    from othello_reinforcement_learning_test_amd.bitboard import OthelloBitboard
    b = OthelloBitboard()
    zero = np.zeros(65, dtype=np.int32)
    p0 = policy_from_visits(zero, b.self_board, b.opp_board, 0.0)
    assert p0.sum() == 1.0 and p0[b.get_legal_moves()[0]] == 1.0
    assert not policy_from_visits(zero, b.self_board, b.opp_board, 1.0).any()
    for val, want in (("0", False), ("false", False), ("", False), ("off", False), ("1", True), ("yes", True)):
        monkeypatch.setenv("OTHELLO_FORCE_DIST", val)
        assert D.force_dist_from_env() is want, val
    monkeypatch.delenv("OTHELLO_FORCE_DIST")
    assert D.force_dist_from_env() is False
