#!/usr/bin/env python3
"""Generate the committed golden vectors by IMPORTING THE REFERENCE in this container.

Run:  python tests/golden/make_golden.py          (needs /root/reference; writes tests/golden/*.npz)

The reference's Python/Cython never enters this repo: only inputs and the outputs it produced
are stored (fixtures G1..G5).  Files are small (each well under 1 MB).

  g1_rules.npz     random reference games: per ply (self, opp, legal, move, flip, terminal, winner)
                   + crafted edge/wrap/pass positions + random unreachable bit patterns
                   + a checksum over 200k LCG-generated positions (legal mask and one flip each)
  g2_tensor.npz    get_tensor_input() bytes for 256 positions
  g3_search.npz    reference MCTS.search / BatchMCTS.search_batch under a closed-form stub
                   evaluator: root visit counts, value sums, returned policy
  g4_net.npz       OthelloResNet forward: weights (small nets) or weight hashes (big nets),
                   inputs, log-probs, values
  g5_episodes.npz  SelfPlayWorker and ParallelSelfPlayWorker episode streams (2x16 net, 5 sims,
                   seeds 42/43): states, pi, z, actions, and the numpy RNG draws they consumed
  g6_arena.npz     src/eval Arena results (greedy/random players, seeded) and GreedyPlayer choices
  g8_extra.npz     OthelloBitboard.get_symmetries of 120 positions with random pi (8 boards + 8 policies each), and
                   MCTS.search at temperatures 0.5 and 2.0 (node.py:175-177's counts ** (1/T)) under the stub evaluator
                   (--only-g8 writes just this file; g7 is tests/golden/make_golden_net6.py)
  g9_terminal.npz  MCTS.search / BatchMCTS.search_batch / get_best_action started AT a terminal position (the root gets
                   the single pass child, mcts.py:76-82 with bitboard.pyx:177-185; every simulation backs the winner up
                   through it): policies at T = 0 / 0.5 / 1 / 2  (--only-g9)
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))  # tests/ (the reference has a `tests` package too)
import build_ref  # noqa: E402

bb = build_ref.import_reference()
from src.mcts.mcts import MCTS  # noqa: E402  (reference)
from src.mcts.node import MCTSNode  # noqa: E402
from src.model.net import OthelloResNet as RefNet  # noqa: E402
from src.train.parallel_self_play import BatchMCTS, ParallelSelfPlayWorker  # noqa: E402
from src.train.self_play import SelfPlayWorker  # noqa: E402

from stub_eval import STUB_LOGITS, stub_logits_values  # noqa: E402

U64 = np.uint64
MASK64 = (1 << 64) - 1


def board_from(s, o):
    b = bb.OthelloBitboard()
    b.self_board = int(s)
    b.opp_board = int(o)
    return b


def flip_of(b, move):
    """Flip mask of a legal move, observed from the reference's own make_move."""
    if move == 64:
        return 0
    c = b.copy()
    before_opp = c.opp_board
    assert c.make_move(move)
    # after the move the sides are swapped: new self == old opp minus flips
    return before_opp & ~c.self_board & MASK64


# ----------------------------------------------------------------------------- G1 / G2
def gen_rules():
    rng = np.random.Generator(np.random.PCG64(20240601))
    rows = []
    game_id = []
    for g in range(220):
        b = bb.OthelloBitboard()
        while True:
            legal_bits = b.get_legal_moves_bits()
            term = b.is_terminal()
            moves = b.get_legal_moves()
            if term:
                rows.append((b.self_board, b.opp_board, legal_bits, 255, 0, 1, b.get_winner(),
                             b.move_count))
                game_id.append(g)
                break
            mv = int(moves[rng.integers(len(moves))])
            fl = flip_of(b, mv)
            rows.append((b.self_board, b.opp_board, legal_bits, mv, fl, 0, b.get_winner(),
                         b.move_count))
            game_id.append(g)
            assert b.make_move(mv)
    arr = np.array([[r[0], r[1], r[2], r[4]] for r in rows], dtype=U64)
    meta = np.array([[r[3], r[5], r[6] & 0xFF, r[7]] for r in rows], dtype=np.int64)

    # crafted positions (SURVEY L2 probes + passes + dead positions)
    crafted = [
        (1 << 0, 1 << 1),                      # self A1, opp B1: C1 must NOT be legal
        (1 << 9, 1 << 8),                      # self B2, opp A2: H1 (7) legal by wrap
        (1 << 7, 1 << 6),                      # self H1, opp G1
        (1 << 63, 1 << 62),
        (1 << 56, 1 << 57),
        ((1 << 28) | (1 << 35), (1 << 27) | (1 << 36)),
        (0xFF, 0xFF00),                        # rows
        (0x0101010101010101, 0x0202020202020202),
        (0x8080808080808080, 0x4040404040404040),
        (0, 0), (MASK64, 0), (0, MASK64),
        (0x00000000000000FF, 0xFFFFFFFFFFFFFF00),
        (0xFFFFFFFFFFFFFF00 & ~(1 << 63), 0xFE),   # nearly full boards
        (0x7FFFFFFFFFFFFFFF, 0),
        (0x5555555555555555, 0xAAAAAAAAAAAAAAAA & ~(1 << 63)),
        (0xAA55AA55AA55AA55 & ~1, 0x55AA55AA55AA55AA & ~(1 << 63)),
    ]
    # random unreachable patterns, varying density
    for i in range(3000):
        dens = rng.uniform(0.05, 0.98)
        occ = 0
        s = 0
        for sq in range(64):
            if rng.random() < dens:
                occ |= 1 << sq
                if rng.random() < 0.5:
                    s |= 1 << sq
        crafted.append((s, occ & ~s))
    c_rows = []
    for s, o in crafted:
        b = board_from(s, o)
        lb = b.get_legal_moves_bits()
        term = b.is_terminal()
        mv = 255
        fl = 0
        if lb:
            ms = [i for i in range(64) if (lb >> i) & 1]
            mv = ms[rng.integers(len(ms))]
            fl = flip_of(b, mv)
        olb = board_from(o, s).get_legal_moves_bits()
        c_rows.append((s, o, lb, fl, mv, int(term), b.get_winner() & 0xFF, olb))
    c_arr = np.array([[r[0], r[1], r[2], r[3], r[7]] for r in c_rows], dtype=U64)
    c_meta = np.array([[r[4], r[5], r[6]] for r in c_rows], dtype=np.int64)

    # invalid-move behaviour: make_move returning False leaves the state unchanged (L4)
    inv = []
    b = bb.OthelloBitboard()
    for mv in [0, 27, 28, 64, -1, 65, 100, 63, 18]:
        c = b.copy()
        ok = c.make_move(mv)
        inv.append((mv, int(ok), c.self_board, c.opp_board, c.move_count, int(c.passed)))

    # checksum over many LCG positions: x_{n+1} = x_n * 6364136223846793005 + 1442695040888963407
    def lcg(x):
        return (x * 6364136223846793005 + 1442695040888963407) & MASK64

    x = 0x9E3779B97F4A7C15
    acc_legal = 0
    acc_flip = 0
    n_ck = 200000
    for i in range(n_ck):
        x = lcg(x); a = x
        x = lcg(x); c = x
        x = lcg(x); d = x
        occ = a | (c & d) if (i & 1) else a & c   # two densities
        s = occ & d
        o = occ & ~d
        bo = board_from(s, o)
        lb = bo.get_legal_moves_bits()
        acc_legal = (acc_legal * 0x100000001B3 + lb) & MASK64
        if lb:
            # flip of the lowest legal move
            mv = (lb & -lb).bit_length() - 1
            fl = flip_of(bo, mv)
            acc_flip = (acc_flip * 0x100000001B3 + fl) & MASK64
    np.savez_compressed(
        os.path.join(HERE, "g1_rules.npz"),
        game_pos=arr, game_meta=meta, game_id=np.array(game_id, dtype=np.int32),
        crafted_pos=c_arr, crafted_meta=c_meta,
        invalid=np.array(inv, dtype=np.int64).astype(np.int64) & np.int64(-1),
        invalid_u64=np.array([[r[2], r[3]] for r in inv], dtype=U64),
        checksum=np.array([n_ck, acc_legal, acc_flip], dtype=U64),
    )
    print("g1: %d game plies, %d crafted, checksum legal=%x flip=%x" %
          (len(rows), len(c_rows), acc_legal, acc_flip))

    # G2 tensors: every 40th game position + first crafted ones
    sel = list(range(0, len(rows), max(1, len(rows) // 200)))[:200]
    pos = [(rows[i][0], rows[i][1]) for i in sel] + crafted[:56]
    t = np.stack([board_from(s, o).get_tensor_input() for s, o in pos])
    assert t.dtype == np.float32 and t.shape[1:] == (3, 8, 8) and t.flags["C_CONTIGUOUS"]
    assert set(np.unique(t)) <= {0.0, 1.0}
    np.savez_compressed(os.path.join(HERE, "g2_tensor.npz"),
                        pos=np.array(pos, dtype=U64), tensor=t.astype(np.uint8))
    print("g2: %d tensors" % len(pos))
    return rows


# ----------------------------------------------------------------------------- G3
class StubModel:
    """Duck-typed `model` for the reference MCTS: closed-form logits/values from the position.

    Logits are drawn from the 16-entry table STUB_LOGITS (tests/stub_eval.py) indexed by a hash of
    (self, opp, action); the value is a dyadic rational in [-1, 1).  The probabilities the
    reference sees are torch.exp(logits): that 16-entry image is recorded in the golden file so
    the checker needs no libm parity.
    """

    def __init__(self):
        self.exp_seen = {}

    def eval(self):
        return self

    def __call__(self, x):
        xn = x.detach().cpu().numpy()
        logits, values, idx = stub_logits_values(xn, return_index=True)
        lt = torch.from_numpy(logits)
        e = torch.exp(lt).numpy()
        for k in range(16):
            m = idx == k
            if m.any():
                vals = np.unique(e[m])
                assert len(vals) == 1, "torch.exp not value-deterministic"
                if k in self.exp_seen:
                    assert self.exp_seen[k] == vals[0]
                self.exp_seen[k] = vals[0]
        return lt, torch.from_numpy(values.reshape(-1, 1))


def root_stats(root):
    n = np.zeros(65, dtype=np.int32)
    w = np.zeros(65, dtype=np.float64)
    p = np.zeros(65, dtype=np.float64)
    for a, ch in root.children.items():
        n[a] = ch.visit_count
        w[a] = ch.value_sum
        p[a] = ch.prior
    return n, w, p


def gen_search(rows):
    stub = StubModel()
    dev = torch.device("cpu")
    rng = np.random.Generator(np.random.PCG64(7))
    cand = [r for r in rows if r[5] == 0]
    # spread over plies, with emphasis on late positions (terminal leaves, passes)
    early = [r for r in cand if r[7] < 40]
    late = [r for r in cand if r[7] >= 40]
    passes = [r for r in cand if r[2] == 0]
    picks = [early[i] for i in rng.choice(len(early), 50, replace=False)]
    picks += [late[i] for i in rng.choice(len(late), 50, replace=False)]
    picks += passes[:10]
    cases = []
    out_n, out_w, out_pi, out_prior = [], [], [], []

    def run_serial(s, o, sims, cp, temp):
        m = MCTS(stub, dev, c_puct=cp)
        # replicate MCTS.search but keep the root (search() discards it): same calls, same order
        board = board_from(s, o)
        root = MCTSNode(prior=1.0)
        pp, _ = m._predict(m._get_board_tensor(board))
        root.expand(pp, board.get_legal_moves())
        for _ in range(sims):
            m._run_simulation(root, board.copy())
        pi = root.get_policy_distribution(temp)
        # and the public entry point must agree with it
        pi2, rv = m.search(board_from(s, o), sims, temperature=temp, add_dirichlet_noise=False)
        assert np.array_equal(pi, pi2) and rv == 0.0
        return root, pi

    for r in picks:
        for sims in (5, 25, 50):
            for cp in (1.0, 1.5):
                root, pi = run_serial(r[0], r[1], sims, cp, 1.0)
                n, w, p = root_stats(root)
                cases.append((r[0], r[1], sims, int(cp * 1000), 0))
                out_n.append(n); out_w.append(w); out_pi.append(pi); out_prior.append(p)
    for r in picks[::5]:
        for sims in (100, 400):
            for cp in (1.0, 1.5):
                root, pi = run_serial(r[0], r[1], sims, cp, 1.0)
                n, w, p = root_stats(root)
                cases.append((r[0], r[1], sims, int(cp * 1000), 0))
                out_n.append(n); out_w.append(w); out_pi.append(pi); out_prior.append(p)
    # temperature-0 policies
    for r in picks[:20]:
        root, pi = run_serial(r[0], r[1], 25, 1.0, 0.0)
        n, w, p = root_stats(root)
        cases.append((r[0], r[1], 25, 1000, 1))
        out_n.append(n); out_w.append(w); out_pi.append(pi); out_prior.append(p)

    # lock-step BatchMCTS must give the same per-game answer (one leaf per game per step)
    bm = BatchMCTS(stub, dev, c_puct=1.0)
    boards = [board_from(r[0], r[1]) for r in picks[:32]]
    res = bm.search_batch(boards, 50, temperature=1.0, add_dirichlet_noise=False)
    batch_pi = np.stack([p for p, _ in res])
    batch_pos = np.array([[r[0], r[1]] for r in picks[:32]], dtype=U64)

    # get_best_action / get_action_evaluations (f3 "next" row; cheap to pin now)
    m = MCTS(stub, dev, c_puct=1.0)
    best, evals = [], []
    for r in picks[:24]:
        best.append(m.get_best_action(board_from(r[0], r[1]), 25))
        evals.append(m.get_action_evaluations(board_from(r[0], r[1]), 25))

    exp_tab = np.array([stub.exp_seen[k] for k in range(16)], dtype=np.float32)
    np.savez_compressed(
        os.path.join(HERE, "g3_search.npz"),
        case_pos=np.array([[c[0], c[1]] for c in cases], dtype=U64),
        case_cfg=np.array([[c[2], c[3], c[4]] for c in cases], dtype=np.int32),
        visits=np.array(out_n, dtype=np.int32), value_sum=np.array(out_w, dtype=np.float64),
        policy=np.array(out_pi, dtype=np.float32), prior=np.array(out_prior, dtype=np.float64),
        batch_pos=batch_pos, batch_pi=batch_pi.astype(np.float32),
        best_pos=np.array([[r[0], r[1]] for r in picks[:24]], dtype=U64),
        best_action=np.array(best, dtype=np.int32), evals=np.array(evals, dtype=np.int32),
        stub_logits=STUB_LOGITS, stub_exp=exp_tab,
    )
    print("g3: %d search cases" % len(cases))


# ----------------------------------------------------------------------------- G4
def sd_hash(sd):
    h = {}
    for k, v in sd.items():
        h[k] = hashlib.sha256(v.detach().cpu().numpy().tobytes()).hexdigest()
    return h


def gen_net(rows):
    rng = np.random.Generator(np.random.PCG64(11))
    cand = [r for r in rows]
    pos = [cand[i] for i in rng.choice(len(cand), 32, replace=False)]
    x = np.stack([board_from(r[0], r[1]).get_tensor_input() for r in pos])
    out = {"pos": np.array([[r[0], r[1]] for r in pos], dtype=U64)}
    for seed in (0, 42):
        for (nb, nf) in ((2, 16), (2, 32), (5, 64), (6, 128), (10, 128)):
            torch.manual_seed(seed)
            net = RefNet(nb, nf).eval()
            if (nb, nf) == (2, 16):
                # give the small net non-trivial BN statistics, as a trained checkpoint has
                g = torch.Generator().manual_seed(100 + seed)
                for mod in net.modules():
                    if isinstance(mod, torch.nn.BatchNorm2d):
                        mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.2)
                        mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                        mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
                        mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            with torch.no_grad():
                logp, v = net(torch.from_numpy(x))
            tag = "s%d_%dx%d" % (seed, nb, nf)
            out[tag + "_logp"] = logp.numpy()
            out[tag + "_v"] = v.numpy()
            sd = net.state_dict()
            if nf <= 32:
                for k, t in sd.items():
                    out[tag + "_sd_" + k] = t.numpy()
            hs = sd_hash(sd)
            out[tag + "_keys"] = np.array(list(hs.keys()))
            out[tag + "_sha"] = np.array(list(hs.values()))
    np.savez_compressed(os.path.join(HERE, "g4_net.npz"), **out)
    print("g4: nets done")


# ----------------------------------------------------------------------------- G5
class RngTap:
    """Records what the reference draws from numpy's global RNG (dirichlet / choice)."""

    def __init__(self):
        self.dir_n = []      # number of legal actions per dirichlet call
        self.choice_a = []   # sampled action per choice call
        self._d = np.random.dirichlet
        self._c = np.random.choice

    def __enter__(self):
        def dirichlet(alpha, size=None):
            self.dir_n.append(len(alpha))
            return self._d(alpha, size)

        def choice(a, size=None, replace=True, p=None):
            r = self._c(a, size, replace, p)
            self.choice_a.append(int(r))
            return r

        np.random.dirichlet = dirichlet
        np.random.choice = choice
        return self

    def __exit__(self, *a):
        np.random.dirichlet = self._d
        np.random.choice = self._c


def make_rec_board(log):
    class RecBoard(bb.OthelloBitboard):
        def make_move(self, a):
            log.append(int(a))
            return super().make_move(a)
    return RecBoard


def gen_episodes():
    out = {}
    for seed in (42, 43):
        for kind in ("serial", "parallel"):
            torch.manual_seed(seed)
            np.random.seed(seed)       # main.py:69-70
            net = RefNet(2, 16).eval()
            log = []
            Rec = make_rec_board(log)
            with RngTap() as tap:
                if kind == "serial":
                    w = SelfPlayWorker(Rec, MCTS(net, torch.device("cpu")), num_simulations=5,
                                       temperature_threshold=10)
                    data = w.execute_episodes(2, add_dirichlet_noise=True)
                else:
                    w = ParallelSelfPlayWorker(Rec, net, torch.device("cpu"), num_simulations=5,
                                               temperature_threshold=10, num_parallel_games=4)
                    data = w.execute_episodes(4, add_dirichlet_noise=True)
            tag = "%s_s%d" % (kind, seed)
            st = np.stack([d[0] for d in data])
            assert set(np.unique(st)) <= {0.0, 1.0}
            out[tag + "_state"] = st.astype(np.uint8)
            out[tag + "_pi"] = np.stack([d[1] for d in data]).astype(np.float32)
            out[tag + "_z"] = np.array([d[2] for d in data], dtype=np.float32)
            out[tag + "_moves"] = np.array(log, dtype=np.int32)
            out[tag + "_dir_n"] = np.array(tap.dir_n, dtype=np.int32)
            out[tag + "_choice"] = np.array(tap.choice_a, dtype=np.int32)
            for k, t in net.state_dict().items():
                out["net_s%d_sd_%s" % (seed, k)] = t.numpy()
            print("g5: %s seed %d: %d samples, %d moves" % (kind, seed, len(data), len(log)))
    np.savez_compressed(os.path.join(HERE, "g5_episodes.npz"), **out)


# ----------------------------------------------------------------------------- G6
def gen_arena():
    """Reference Arena + host players (src/eval): deterministic and seeded matches, result fields only."""
    import random as pyrandom

    from src.eval.arena import Arena
    from src.eval.players import GreedyPlayer, RandomPlayer
    out = {}
    arena = Arena(verbose=False)
    res = arena.play_matches(GreedyPlayer("G1"), GreedyPlayer("G2"), num_games=4, alternate_colors=True)
    out["greedy_greedy"] = np.array([[r.winner, r.player1_score, r.player2_score, r.num_moves] for r in res],
                                    dtype=np.int32)
    for seed in (1, 2):
        pyrandom.seed(seed)
        res = arena.play_matches(RandomPlayer("R"), GreedyPlayer("G"), num_games=12, alternate_colors=True)
        out["random_greedy_s%d" % seed] = np.array(
            [[r.winner, r.player1_score, r.player2_score, r.num_moves] for r in res], dtype=np.int32)
        pyrandom.seed(seed)
        res = arena.play_matches(GreedyPlayer("G"), RandomPlayer("R"), num_games=6, alternate_colors=False)
        out["greedy_random_s%d" % seed] = np.array(
            [[r.winner, r.player1_score, r.player2_score, r.num_moves] for r in res], dtype=np.int32)
    # greedy move choice on sampled positions
    rng = np.random.Generator(np.random.PCG64(4))
    g = GreedyPlayer()
    pos, act = [], []
    for _ in range(6):
        b = bb.OthelloBitboard()
        while not b.is_terminal():
            pos.append((b.self_board, b.opp_board, b.move_count))
            act.append(g.get_action(b))
            mv = b.get_legal_moves()
            b.make_move(int(mv[rng.integers(len(mv))]))
    out["greedy_pos"] = np.array([[p[0], p[1]] for p in pos], dtype=U64)
    out["greedy_mc"] = np.array([p[2] for p in pos], dtype=np.int32)
    out["greedy_action"] = np.array(act, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "g6_arena.npz"), **out)
    print("g6: arena results and %d greedy choices" % len(act))


def gen_extra():
    """g8: symmetries (bitboard.pyx:338-370) and general-temperature policies (node.py:162-182)."""
    g1 = np.load(os.path.join(HERE, "g1_rules.npz"))
    rng = np.random.Generator(np.random.PCG64(88))
    pos = g1["game_pos"]
    idx = rng.choice(len(pos), 120, replace=False)
    sym_pos, sym_pi, sym_states, sym_pis = [], [], [], []
    for i in idx:
        s, o = int(pos[i, 0]), int(pos[i, 1])
        pi = rng.dirichlet([0.3] * 65).astype(np.float32)
        out = board_from(s, o).get_symmetries(pi)
        assert len(out) == 8
        sym_pos.append((s, o)); sym_pi.append(pi)
        sym_states.append(np.stack([np.asarray(b, dtype=np.float32) for b, _ in out]))
        sym_pis.append(np.stack([np.asarray(p, dtype=np.float32) for _, p in out]))
    stub = StubModel()
    dev = torch.device("cpu")
    cases, pol = [], []
    for i in rng.choice(len(pos), 40, replace=False):
        s, o = int(pos[i, 0]), int(pos[i, 1])
        b = board_from(s, o)
        if b.is_terminal():
            continue
        for temp in (0.5, 2.0):
            for sims in (25, 50):
                m = MCTS(stub, dev, c_puct=1.0)
                pi, rv = m.search(board_from(s, o), sims, temperature=temp, add_dirichlet_noise=False)
                assert rv == 0.0 and pi.dtype == np.float32
                cases.append((s, o, sims, int(temp * 1000)))
                pol.append(pi)
    g3_exp = np.load(os.path.join(HERE, "g3_search.npz"))["stub_exp"]   # the checker uses g3's table of torch.exp
    for k, v in stub.exp_seen.items():
        assert np.float32(v) == g3_exp[k]
    np.savez_compressed(
        os.path.join(HERE, "g8_extra.npz"),
        sym_pos=np.array(sym_pos, dtype=U64), sym_pi=np.array(sym_pi, dtype=np.float32),
        sym_states=np.array(sym_states, dtype=np.float32), sym_pis=np.array(sym_pis, dtype=np.float32),
        temp_pos=np.array([[c[0], c[1]] for c in cases], dtype=U64),
        temp_cfg=np.array([[c[2], c[3]] for c in cases], dtype=np.int32),
        temp_policy=np.array(pol, dtype=np.float32), stub_logits=STUB_LOGITS,
    )
    print("g8: %d symmetry cases, %d temperature cases" % (len(sym_pos), len(cases)))


def gen_terminal():
    """g9: searches whose ROOT is terminal (round-3 advisor finding: no fixture covered it)."""
    rng = np.random.Generator(np.random.PCG64(99))
    stub = StubModel()
    dev = torch.device("cpu")
    finals = []
    while len(finals) < 24:
        b = bb.OthelloBitboard()
        b.reset()
        while not b.is_terminal():
            mv = b.get_legal_moves()
            assert b.make_move(int(mv[rng.integers(len(mv))]))
        if (int(b.self_board), int(b.opp_board)) not in finals:
            finals.append((int(b.self_board), int(b.opp_board)))
    cases, pol = [], []
    for s, o in finals:
        for temp in (0.0, 0.5, 1.0, 2.0):
            for sims in (1, 10):
                m = MCTS(stub, dev, c_puct=1.0)
                pi, rv = m.search(board_from(s, o), sims, temperature=temp, add_dirichlet_noise=False)
                assert rv == 0.0 and pi.dtype == np.float32
                cases.append((s, o, sims, int(temp * 1000)))
                pol.append(pi)
    bm = BatchMCTS(stub, dev, c_puct=1.0)
    res = bm.search_batch([board_from(s, o) for s, o in finals], 8, temperature=1.0, add_dirichlet_noise=False)
    batch_pi = np.stack([p for p, _ in res])
    m = MCTS(stub, dev, c_puct=1.0)
    best = [m.get_best_action(board_from(s, o), 5) for s, o in finals]
    np.savez_compressed(
        os.path.join(HERE, "g9_terminal.npz"),
        pos=np.array(finals, dtype=U64),
        case_pos=np.array([[c[0], c[1]] for c in cases], dtype=U64),
        case_cfg=np.array([[c[2], c[3]] for c in cases], dtype=np.int32),
        policy=np.array(pol, dtype=np.float32), batch_pi=batch_pi.astype(np.float32),
        best_action=np.array(best, dtype=np.int32), stub_logits=STUB_LOGITS,
    )
    print("g9: %d terminal roots, %d cases; pi[64] values %s; best %s"
          % (len(finals), len(cases), np.unique(np.array(pol)[:, 64]), np.unique(best)))


if __name__ == "__main__":
    torch.set_num_threads(1)
    if "--only-g9" in sys.argv:
        gen_terminal()
        sys.exit(0)
    if "--only-arena" in sys.argv:
        gen_arena()
        sys.exit(0)
    if "--only-g8" in sys.argv:
        gen_extra()
        sys.exit(0)
    rows = gen_rules()
    gen_search(rows)
    gen_net(rows)
    gen_episodes()
    gen_arena()
    gen_extra()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-18s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))
