"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Bit-exact for rules, search statistics and replay tuples; 1e-4 for the network
(BASELINE.json north_star).  Run on the MI355X box with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_lib as ol
from stub_eval import stub_probs_values

pytestmark = pytest.mark.gpu

U64 = np.uint64


@pytest.fixture(scope="module")
def pkg():
    import othello_reinforcement_learning_test_amd as p
    p._lib.require_device()   # fail loudly, never skip: -m gpu means a GPU box
    return p


def dev_u64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=U64).view(np.int64)).cuda()


def host_u64(t):
    return t.cpu().numpy().view(U64)


def random_positions(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    a, c, d = (rng.integers(0, 2**64, n, dtype=U64) for _ in range(3))
    occ = np.where(np.arange(n) % 2 == 0, a & c, a | (c & d))
    return occ & d, occ & ~d


def game_positions(n_games, seed):
    """Reachable positions (with their ply) from random playouts on the oracle."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n_games):
        b = ol.board()
        while not ol.lib().orc_is_terminal(b):
            out.append((b.self_board, b.opp_board, b.move_count))
            mv = ol.legal_list(b)
            ol.lib().orc_make_move(b, int(mv[rng.integers(len(mv))]))
    return out


# =========================================================================================== rules
def test_rules_batch_vs_golden_and_oracle(pkg, golden):
    DB = pkg.DeviceBoards
    g = golden("g1_rules.npz")
    for arr in (g["game_pos"], g["crafted_pos"]):
        s, o = arr[:, 0], arr[:, 1]
        assert np.array_equal(host_u64(DB.legal_moves(dev_u64(s), dev_u64(o))), arr[:, 2])
    s, o = random_positions(200_000, 1)
    assert np.array_equal(host_u64(DB.legal_moves(dev_u64(s), dev_u64(o))), ol.legal_batch(s, o))
    term, win = DB.status(dev_u64(s[:20000]), dev_u64(o[:20000]))
    for i in range(0, 20000, 97):
        b = ol.board(s[i], o[i])
        assert term[i].item() == ol.lib().orc_is_terminal(b) and win[i].item() == ol.lib().orc_winner(b)


def test_make_move_batch(pkg, golden):
    DB = pkg.DeviceBoards
    g = golden("g1_rules.npz")
    pos, meta, gid = g["game_pos"], g["game_meta"], g["game_id"]
    nt = meta[:, 1] == 0
    s, o, mv = pos[nt, 0], pos[nt, 1], meta[nt, 0].astype(np.int32)
    ds, do = dev_u64(s), dev_u64(o)
    ok, flips = DB.make_move(ds, do, torch.from_numpy(mv).cuda())
    assert ok.cpu().numpy().all()
    real = mv < 64
    assert np.array_equal(host_u64(flips)[real], pos[nt, 3][real])
    idx = np.nonzero(nt)[0]
    same = gid[idx + 1] == gid[idx]
    assert np.array_equal(host_u64(ds)[same], pos[idx + 1, 0][same])
    assert np.array_equal(host_u64(do)[same], pos[idx + 1, 1][same])
    # invalid moves: occupied squares, no-flip squares, pass with moves available, out of range
    s2, o2 = random_positions(50_000, 2)
    rng = np.random.Generator(np.random.PCG64(8))
    mv2 = rng.integers(-2, 67, len(s2)).astype(np.int32)
    ds, do = dev_u64(s2), dev_u64(o2)
    ok, flips = DB.make_move(ds, do, torch.from_numpy(mv2).cuda())
    ok = ok.cpu().numpy()
    hs, ho = host_u64(ds), host_u64(do)
    for i in range(0, len(s2), 37):
        b = ol.board(s2[i], o2[i])
        r = ol.lib().orc_make_move(b, int(mv2[i]))
        assert r == ok[i] and (b.self_board, b.opp_board) == (hs[i], ho[i])


def test_tensor_batch(pkg, golden):
    g = golden("g2_tensor.npz")
    t = pkg.DeviceBoards.tensor_input(dev_u64(g["pos"][:, 0]), dev_u64(g["pos"][:, 1]))
    assert t.dtype == torch.float32 and tuple(t.shape[1:]) == (3, 8, 8)
    assert np.array_equal(t.cpu().numpy(), g["tensor"].astype(np.float32))


def test_rules_checksum_full_size(pkg, golden):
    """Size-independent property: checksum of legal masks / flips over an LCG position stream."""
    n, la, fa = (int(x) for x in golden("g1_rules.npz")["checksum"])
    a, b = C.c_uint64(0), C.c_uint64(0)
    pkg._lib.call("oth_rules_checksum", n, C.byref(a), C.byref(b), None)
    assert (a.value, b.value) == (la, fa)          # against the reference-generated value
    big = 3_000_000
    pkg._lib.call("oth_rules_checksum", big, C.byref(a), C.byref(b), None)
    assert (a.value, b.value) == ol.rules_checksum(big)   # against the oracle at a larger size


# =========================================================================================== search
@pytest.fixture(scope="module")
def g3(golden):
    return golden("g3_search.npz")


def test_search_golden_cases(pkg, g3):
    """Root visit counts, float64 value sums, float32 priors and pi equal the reference's, bit for
    bit, under the closed-form stub evaluator."""
    table = g3["stub_exp"]
    cfg = g3["case_cfg"]
    fn = lambda s, o, lg: stub_probs_values(s, o, table)   # noqa: E731
    for sims in sorted(set(cfg[:, 0])):
        for cp in sorted(set(cfg[:, 1])):
            for t0 in (0, 1):
                sel = np.nonzero((cfg[:, 0] == sims) & (cfg[:, 1] == cp) & (cfg[:, 2] == t0))[0]
                if len(sel) == 0:
                    continue
                eng = pkg.SearchEngine(len(sel), int(sims), c_puct=cp / 1000.0)
                pos = g3["case_pos"][sel]
                pi, visits, wsum, prior = eng.search_with(pos[:, 0], pos[:, 1], fn, 0.0 if t0 else 1.0)
                assert np.array_equal(visits, g3["visits"][sel]), (sims, cp)
                assert np.array_equal(wsum, g3["value_sum"][sel])
                assert np.array_equal(prior, g3["prior"][sel].astype(np.float32))
                assert np.array_equal(pi, g3["policy"][sel])
                assert eng.counters()["simulations"] == len(sel) * int(sims)


def test_search_general_temperature_vs_reference(pkg, g3, golden):
    """MCTS.search / get_action_probs with temperature 0.5 and 2.0 (node.py:175-177: counts ** (1/T), renormalised in
    float32) == the reference's own answers (g8, generated by the reference under the stub evaluator), through both
    SearchEngine.search_with and BatchMCTS.search_batch; the round-2 ValueError for T not in {0, 1} is gone."""
    g8 = golden("g8_extra.npz")
    table = g3["stub_exp"]
    fn = lambda s, o, lg: stub_probs_values(s, o, table)   # noqa: E731
    cfg = g8["temp_cfg"]
    for sims in sorted(set(cfg[:, 0])):
        for t1000 in sorted(set(cfg[:, 1])):
            sel = np.nonzero((cfg[:, 0] == sims) & (cfg[:, 1] == t1000))[0]
            eng = pkg.SearchEngine(len(sel), int(sims), c_puct=1.0)
            pos = g8["temp_pos"][sel]
            pi, _, _, _ = eng.search_with(pos[:, 0], pos[:, 1], fn, t1000 / 1000.0)
            assert np.array_equal(pi, g8["temp_policy"][sel]), (sims, t1000)

    class Stub:
        def host_eval(self, s, o, lg):
            return stub_probs_values(s, o, table)
    bm = pkg.BatchMCTS(None, evaluator=Stub(), c_puct=1.0)
    sel = np.nonzero((cfg[:, 0] == 25) & (cfg[:, 1] == 500))[0][:9]
    boards = []
    for s, o in g8["temp_pos"][sel]:
        b = pkg.OthelloBitboard()
        b.self_board, b.opp_board = int(s), int(o)
        boards.append(b)
    res = bm.search_batch(boards, 25, temperature=0.5)
    assert np.array_equal(np.stack([p for p, _ in res]), g8["temp_policy"][sel])


def test_search_at_terminal_root_vs_reference(pkg, g3, golden):
    """Searches whose ROOT is terminal (g9, reference-generated): pass child only, pi one-hot on 64 at T = 0 / 0.5 / 1 /
    2, through SearchEngine.search_with (device k_results for T in {0, 1}, host policy_from_visits otherwise),
    BatchMCTS.search_batch and get_best_action's reduction of the T = 0 policy."""
    g9 = golden("g9_terminal.npz")
    table = g3["stub_exp"]
    fn = lambda s, o, lg: stub_probs_values(s, o, table)   # noqa: E731
    cfg = g9["case_cfg"]
    for sims in sorted(set(cfg[:, 0])):
        for t1000 in sorted(set(cfg[:, 1])):
            sel = np.nonzero((cfg[:, 0] == sims) & (cfg[:, 1] == t1000))[0]
            eng = pkg.SearchEngine(len(sel), int(sims), c_puct=1.0)
            pos = g9["case_pos"][sel]
            pi, visits, _, _ = eng.search_with(pos[:, 0], pos[:, 1], fn, t1000 / 1000.0)
            assert np.array_equal(pi, g9["policy"][sel]), (sims, t1000)
            assert (visits[:, 64] == sims).all() and visits[:, :64].sum() == 0

    class Stub:
        def host_eval(self, s, o, lg):
            return stub_probs_values(s, o, table)
    boards = []
    for s, o in g9["pos"]:
        b = pkg.OthelloBitboard()
        b.self_board, b.opp_board = int(s), int(o)
        boards.append(b)
    bm = pkg.BatchMCTS(None, evaluator=Stub(), c_puct=1.0)
    res = bm.search_batch(boards, 8, temperature=1.0)
    assert np.array_equal(np.stack([p for p, _ in res]), g9["batch_pi"])
    from othello_reinforcement_learning_test_amd.mcts import best_action_from_policy
    eng = pkg.SearchEngine(len(boards), 5, c_puct=1.0)
    pi0, _, _, _ = eng.search_with(g9["pos"][:, 0], g9["pos"][:, 1], fn, temperature=0.0)
    assert [best_action_from_policy(pi0[i], b.get_legal_moves()) for i, b in enumerate(boards)] \
        == [int(a) for a in g9["best_action"]]


def test_search_vs_oracle_many_positions(pkg, g3):
    """Fresh seeded positions, including late-game ones with terminal leaves and forced passes."""
    table = g3["stub_exp"]
    ev = ol.make_eval(lambda s, o: stub_probs_values(s, o, table))
    pos = game_positions(12, 77)
    late = [p for p in pos if p[2] >= 50][:96]
    early = pos[:160:2]
    pick = early + late
    eng = pkg.SearchEngine(len(pick), 50, c_puct=1.0)
    pi, visits, wsum, prior = eng.search_with([p[0] for p in pick], [p[1] for p in pick],
                                              lambda s, o, lg: stub_probs_values(s, o, table))
    assert eng.counters()["terminal_sims"] > 0
    for i, (s, o, _) in enumerate(pick):
        opi, on, ow, opr = ol.search(ol.board(s, o), 50, 1.0, 1.0, ev)
        assert np.array_equal(visits[i], on), i
        assert np.array_equal(wsum[i], ow) and np.array_equal(pi[i], opi)
        assert np.array_equal(prior[i], opr.astype(np.float32))


def test_search_deep_tree_400_sims(pkg, g3):
    table = g3["stub_exp"]
    ev = ol.make_eval(lambda s, o: stub_probs_values(s, o, table))
    pick = game_positions(2, 5)[10:110:7]
    eng = pkg.SearchEngine(len(pick), 400, c_puct=1.5)
    pi, visits, wsum, _ = eng.search_with([p[0] for p in pick], [p[1] for p in pick],
                                          lambda s, o, lg: stub_probs_values(s, o, table))
    for i, (s, o, _) in enumerate(pick):
        _, on, ow, _ = ol.search(ol.board(s, o), 400, 1.5, 1.0, ev)
        assert np.array_equal(visits[i], on) and np.array_equal(wsum[i], ow)


def test_search_maximum_simulations(pkg, g3):
    """Largest search the engine accepts (oth_engine_cfg.num_simulations = 4000: the descent path of a workgroup's
    four games just fits 64 KiB of LDS, visit counters are u16, child links u16): still bit-exact against the oracle,
    including an end-game position whose tree saturates with terminal leaves.  4001 is rejected."""
    table = g3["stub_exp"]
    ev = ol.make_eval(lambda s, o: stub_probs_values(s, o, table))
    pos = game_positions(3, 9)
    pick = [pos[3], pos[40], [p for p in pos if p[2] >= 52][0]]
    eng = pkg.SearchEngine(len(pick), 4000, c_puct=1.0)
    pi, visits, wsum, _ = eng.search_with([p[0] for p in pick], [p[1] for p in pick],
                                          lambda s, o, lg: stub_probs_values(s, o, table))
    assert visits.sum(axis=1).tolist() == [4000] * len(pick) and visits.max() > 1000
    for i, (s, o, _) in enumerate(pick):
        opi, on, ow, _ = ol.search(ol.board(s, o), 4000, 1.0, 1.0, ev)
        assert np.array_equal(visits[i], on) and np.array_equal(wsum[i], ow) and np.array_equal(pi[i], opi)
    with pytest.raises(Exception):
        pkg.SearchEngine(1, 4001)


def test_best_action_and_evaluations_vs_reference(pkg, g3):
    """get_best_action / get_action_evaluations (reference mcts.py:257-362) from the device search statistics,
    against the reference's answers under the stub evaluator."""
    from othello_reinforcement_learning_test_amd.mcts import best_action_from_policy, evaluations_from_stats
    table = g3["stub_exp"]
    fn = lambda s, o, lg: stub_probs_values(s, o, table)   # noqa: E731
    pos = g3["best_pos"]
    eng = pkg.SearchEngine(len(pos), 25, c_puct=1.0)
    pi0, _, _, _ = eng.search_with(pos[:, 0], pos[:, 1], fn, temperature=0.0)
    _, visits, wsum, _ = eng.search_with(pos[:, 0], pos[:, 1], fn, temperature=1.0)
    for i, (s, o) in enumerate(pos):
        legal = ol.legal_list(ol.board(s, o))
        assert best_action_from_policy(pi0[i], legal) == g3["best_action"][i]
        assert np.array_equal(evaluations_from_stats(visits[i], wsum[i], legal), g3["evals"][i])


def test_search_api_properties(pkg):
    """Restated from the reference's tests/test_mcts.py:158-184,236-256: pi has 65 entries, sums to
    one, is zero off the legal moves; T=0 gives exactly one non-zero entry; best action is legal."""
    torch.manual_seed(0)
    net = pkg.OthelloResNet(2, 16).eval()
    m = pkg.MCTS(net, torch.device("cuda"))
    b = pkg.OthelloBitboard()
    pi, rv = m.search(b, 10, temperature=1.0)
    assert pi.shape == (65,) and pi.dtype == np.float32 and abs(pi.sum() - 1) < 1e-6 and rv == 0.0
    legal = b.get_legal_moves()
    assert all(pi[a] == 0 for a in range(65) if a not in legal)
    pi0, _ = m.search(b, 10, temperature=0.0)
    assert (pi0 != 0).sum() == 1
    assert m.get_best_action(b, 10) in legal
    ev = m.get_action_evaluations(b, 10)
    assert ev.dtype == np.int32 and ev.shape == (65,) and ev.min() >= 0 and ev.max() <= 100


# =========================================================================================== network
NETS = [(2, 16), (2, 32), (5, 64), (6, 128), (10, 128)]


def _golden_net(pkg, g, seed, nb, nf):
    tag = "s%d_%dx%d" % (seed, nb, nf)
    torch.manual_seed(seed)
    net = pkg.OthelloResNet(nb, nf).eval()
    if (nb, nf) == (2, 16):
        net.load_state_dict({k: torch.from_numpy(g[tag + "_sd_" + k]) for k in net.state_dict()})
    return tag, net


@pytest.mark.parametrize("seed", [0, 42])
def test_net_forward_vs_reference_outputs(pkg, golden, seed):
    """HIP forward vs the reference's own outputs (golden g4), tolerance 1e-4 (north_star)."""
    g = golden("g4_net.npz")
    x = np.stack([ol.tensor(ol.board(s, o)) for s, o in g["pos"]])
    xd = torch.from_numpy(x).cuda()
    for nb, nf in NETS:
        tag, net = _golden_net(pkg, g, seed, nb, nf)
        # 128 filters: "f16x3" is the Winograd trunk (k_trunk_w), "f16x3_direct" the direct-convolution one (k_trunk16)
        precs = ["f32"] + (["f16x3"] if nf in (32, 64, 128) else []) + (["f16x3_direct"] if nf == 128 else [])
        for prec in precs:
            ev = pkg.HipResNetEvaluator(net, precision=prec)
            logp, v = ev.forward_planes(xd)
            e1 = np.abs(logp.cpu().numpy() - g[tag + "_logp"]).max()
            e2 = np.abs(v.cpu().numpy() - g[tag + "_v"]).max()
            assert e1 < 1e-4 and e2 < 1e-4, (tag, prec, e1, e2)
            assert np.allclose(np.exp(logp.cpu().numpy()).sum(1), 1.0, atol=1e-5)  # test_model.py:63-75


NETS6 = [(2, 16), (2, 32), (5, 64), (3, 128)]


@pytest.mark.parametrize("seed", [0, 42])
def test_net6_forward_vs_reference_outputs(pkg, golden, seed):
    """6x6 networks (BASELINE.json configs[4]; reference configs/debug_6x6.yaml = 5x64 on 6x6): the exact-fp32 MFMA
    trunk vs the reference's own (N,37) / (N,1) outputs (g7), tolerance 1e-4."""
    g = golden("g7_net6.npz")
    xd = torch.from_numpy(g["x"]).cuda()
    for nb, nf in NETS6:
        tag = "s%d_%dx%d" % (seed, nb, nf)
        torch.manual_seed(seed)
        net = pkg.OthelloResNet(nb, nf, board_size=6).eval()
        if (nb, nf) == (2, 16):
            net.load_state_dict({k: torch.from_numpy(g[tag + "_sd_" + k]) for k in net.state_dict()})
        for prec in ["f32"] + (["f16x3"] if nf in (32, 64, 128) else []):
            ev = pkg.HipResNetEvaluator(net, precision=prec)
            assert ev.policy_size == 37
            logp, v = ev.forward_planes(xd)
            assert tuple(logp.shape) == (len(xd), 37)
            e1 = np.abs(logp.cpu().numpy() - g[tag + "_logp"]).max()
            e2 = np.abs(v.cpu().numpy() - g[tag + "_v"]).max()
            assert e1 < 1e-4 and e2 < 1e-4, (tag, prec, e1, e2)
        assert pkg.HipResNetEvaluator(net).precision == ("f16x3" if nf in (32, 64, 128) else "f32")


@pytest.mark.parametrize("nb,nf,bs,prec", [(2, 16, 8, "f32"), (2, 32, 8, "f32"), (5, 64, 8, "f32"), (2, 128, 8, "f32"),
                                           (5, 64, 6, "f32"), (2, 16, 6, "f32"), (2, 32, 6, "f32"), (2, 128, 6, "f32"),
                                           (5, 64, 8, "f16x3"), (3, 32, 8, "f16x3"), (5, 64, 6, "f16x3"),
                                           (3, 32, 6, "f16x3"), (2, 128, 6, "f16x3")])
def test_wave_trunks_ragged_batches(pkg, nb, nf, bs, prec):
    """The wave-per-position trunks -- exact fp32 MFMA (net_f32.hip) and the fp16-split one for 32 / 64 filters
    (net_h3.hip) -- on trained-like weights at batch sizes that leave waves, workgroups and tiles partly empty, plus a
    3000-position batch, vs torch fp32 on the same weights (1e-4); and a device-side batch length (n_valid)."""
    torch.manual_seed(1000 + nf + bs)
    net = pkg.OthelloResNet(nb, nf, board_size=bs).eval()
    g = torch.Generator().manual_seed(7)
    for mod in net.modules():   # trained-like BatchNorm statistics
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.2)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    ev = pkg.HipResNetEvaluator(net, precision=prec)
    rng = np.random.Generator(np.random.PCG64(5))
    for n in (1, 2, 3, 5, 7, 8, 9, 17, 33, 3000):
        occ = rng.random((n, bs, bs)) < 0.6
        own = occ & (rng.random((n, bs, bs)) < 0.5)
        x = np.stack([own, occ & ~own, (~occ) & (rng.random((n, bs, bs)) < 0.5)], axis=1).astype(np.float32)
        logp, v = ev.forward_planes(torch.from_numpy(x).cuda())
        with torch.no_grad():
            rl, rv = net(torch.from_numpy(x))
        e1 = (logp.cpu() - rl).abs().max().item()
        e2 = (v.cpu() - rv).abs().max().item()
        assert e1 < 1e-4 and e2 < 1e-4, (n, e1, e2)
    # device-side batch length: rows beyond *n_valid are not evaluated (left untouched)
    n, nvalid = 64, 37
    w = (np.uint64(1) << np.arange(bs * bs, dtype=U64))
    sb = (x[:n, 0].reshape(n, -1).astype(U64) * w).sum(1, dtype=U64)
    ob = (x[:n, 1].reshape(n, -1).astype(U64) * w).sum(1, dtype=U64)
    lg = (x[:n, 2].reshape(n, -1).astype(U64) * w).sum(1, dtype=U64)
    logp2 = torch.full((n, bs * bs + 1), 7.0, device="cuda")
    v2 = torch.full((n,), 7.0, device="cuda")
    nv = torch.tensor([nvalid], dtype=torch.int32, device="cuda")
    dsb, dob, dlg = dev_u64(sb), dev_u64(ob), dev_u64(lg)
    pkg._lib.call("oth_net_forward_bits", ev.handle, dsb.data_ptr(), dob.data_ptr(), dlg.data_ptr(), n,
                  nv.data_ptr(), logp2.data_ptr(), v2.data_ptr(), pkg._lib.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(logp2[:nvalid], logp[:nvalid]) and torch.equal(v2[:nvalid], v[:nvalid, 0])
    assert bool((logp2[nvalid:] == 7.0).all()) and bool((v2[nvalid:] == 7.0).all())


@pytest.mark.parametrize("nb,nf,bs,prec", [(2, 32, 8, "f16x3"), (2, 32, 6, "f16x3"), (5, 64, 8, "f16x3"), (5, 64, 6, "f16x3"),
                                           (2, 16, 8, "f32"), (2, 16, 6, "f32"), (2, 128, 6, "f16x3"),
                                           (2, 128, 8, "f16x3"), (2, 128, 8, "f16x3_direct"), (2, 128, 8, "f32"),
                                           (2, 32, 8, "f32"), (2, 32, 6, "f32"), (5, 64, 8, "f32"), (5, 64, 6, "f32")])
def test_wave_trunks_full_occupancy_reproducible(pkg, nb, nf, bs, prec):
    """4096 positions per launch -- several workgroups resident per CU -- five times over: every output bit-identical from
    launch to launch and within 1e-4 of torch fp32.  (Round 3: a batched form of the value head's first FC gave wrong
    values at ~1 % of the positions of the 32-filter 8x8 kernel, only with two workgroups per CU and differently in every
    launch; smaller batches and a single launch compared at 1e-4 on 3000 positions did not catch it.)"""
    torch.manual_seed(42)
    net = pkg.OthelloResNet(nb, nf, board_size=bs).eval()
    rng = np.random.Generator(np.random.PCG64(11))
    n = 4096
    occ = rng.random((n, bs, bs)) < 0.5
    own = occ & (rng.random((n, bs, bs)) < 0.5)
    x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((n, bs, bs)) < 0.4)], 1).astype(np.float32)).cuda()
    with torch.no_grad():
        rl, rv = net.cuda()(x)
    ev = pkg.HipResNetEvaluator(net.cpu(), precision=prec)
    logp0, v0 = ev.forward_planes(x)
    torch.cuda.synchronize()
    logp0, v0 = logp0.clone(), v0.clone()
    assert (logp0 - rl).abs().max().item() < 1e-4 and (v0 - rv).abs().max().item() < 1e-4
    for _ in range(4):
        logp, v = ev.forward_planes(x)
        torch.cuda.synchronize()
        assert torch.equal(logp, logp0), "policy outputs differ between launches"
        assert torch.equal(v, v0), ("value outputs differ between launches", int((v != v0).sum().item()))


def test_net_large_batch_and_ragged_tail(pkg):
    """4096+3 real positions on the 10x128 network: fp32-equivalent MFMA trunk vs torch fp32 on the
    same weights (tolerance 1e-4), every row; batch sizes that are not a multiple of the tile."""
    torch.manual_seed(42)
    net = pkg.OthelloResNet(10, 128).eval()
    pos = game_positions(80, 3)
    rng = np.random.Generator(np.random.PCG64(0))
    idx = rng.choice(len(pos), 4099, replace=True)
    s = np.array([pos[i][0] for i in idx], dtype=U64)
    o = np.array([pos[i][1] for i in idx], dtype=U64)
    ds, do = dev_u64(s), dev_u64(o)
    lg = pkg.DeviceBoards.legal_moves(ds, do)
    x = pkg.DeviceBoards.tensor_input(ds, do)
    ev = pkg.HipResNetEvaluator(net)       # default precision for 128 filters: f16x3
    assert ev.precision == "f16x3"
    logp, v = ev.forward_bits(ds, do, lg)
    netd = net.cuda()
    with torch.no_grad():
        rl, rv = netd(x)
    assert (logp - rl).abs().max().item() < 1e-4 and (v - rv).abs().max().item() < 1e-4
    for n in (1, 2, 3, 5, 63, 257):
        l2, v2 = ev.forward_bits(ds[:n].contiguous(), do[:n].contiguous(), lg[:n].contiguous())
        assert torch.equal(l2, logp[:n]) and torch.equal(v2, v[:n])   # per-row results independent of batch


def test_trunk_kernel_variants_agree(pkg):
    """All builds of the fused 128-filter trunk -- the direct kernel k_trunk16 with one or two positions per workgroup
    (fp16x3) and four (the single-pass f16 precision), the Winograd trunk k_trunk_w with one or two -- give the same answer
    to ~1e-6 and stay within 1e-4 of torch fp32; the one- and two-position builds are picked by environment variables read
    at launch time.  (The 32x32x16 first version of the direct kernel and its four-position fp16x3 build were removed from
    the sources in round 5.)"""
    import os
    torch.manual_seed(7)
    net = pkg.OthelloResNet(6, 128).eval()
    pos = game_positions(6, 11)[:301]
    s = np.array([p[0] for p in pos], dtype=U64)
    o = np.array([p[1] for p in pos], dtype=U64)
    ds, do = dev_u64(s), dev_u64(o)
    lg = pkg.DeviceBoards.legal_moves(ds, do)
    with torch.no_grad():
        rl, rv = net.cuda()(pkg.DeviceBoards.tensor_input(ds, do))
    net.cpu()
    outs = []
    old = {k: os.environ.get(k) for k in ("OTH_TRUNK_TP", "OTH_WINO_TP")}
    try:
        direct = []
        for tp in ("2", "1"):   # the direct-convolution builds: bit-identical to each other
            os.environ["OTH_TRUNK_TP"] = tp
            ev = pkg.HipResNetEvaluator(net, precision="f16x3_direct")
            logp, v = ev.forward_bits(ds, do, lg)
            assert (logp - rl).abs().max().item() < 1e-4 and (v - rv).abs().max().item() < 1e-4, ("direct", tp)
            direct.append((logp, v))
            outs.append(logp)
        assert torch.equal(direct[0][0], direct[1][0]) and torch.equal(direct[0][1], direct[1][1])
        os.environ.pop("OTH_TRUNK_TP", None)
        ev = pkg.HipResNetEvaluator(net, precision="f16")     # four positions per workgroup; a single f16 pass is NOT
        logp, v = ev.forward_bits(ds, do, lg)                   # parity-grade: 2e-3 here
        assert (logp - rl).abs().max().item() < 2e-3 and (v - rv).abs().max().item() < 2e-3
        wino = []
        for tp in ("1", "2"):   # the Winograd trunk, one- and two-position builds: bit-identical to each other
            os.environ["OTH_WINO_TP"] = tp
            ev = pkg.HipResNetEvaluator(net, precision="f16x3")
            logp, v = ev.forward_bits(ds, do, lg)
            assert ev.kernel_info(len(s))["kernel"].startswith("k_trunk_w<%s>" % tp)     # the library names what it runs
            assert (logp - rl).abs().max().item() < 1e-4 and (v - rv).abs().max().item() < 1e-4, ("wino", tp)
            wino.append((logp, v))
            outs.append(logp)
        assert torch.equal(wino[0][0], wino[1][0]) and torch.equal(wino[0][1], wino[1][1])
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
    for x in outs[1:]:
        assert (x - outs[0]).abs().max().item() < 5e-6


def test_wino6_variants_agree(pkg):
    """BASELINE configs[4]'s network (5 x 64 on 6x6) through its two trunk builds -- k_trunk_h3 (OTH_WINO6=0, direct 3x3) and
    the Winograd trunk k_trunk_w6 (default): each within 1e-4 of torch fp32 and within 1e-5 of each other, a position's
    outputs bit-identical whatever the batch it is evaluated in; ragged batch sizes included.  (The eight-wave experiment
    k_trunk_w6b of round 3 was removed in round 4 with the V image it was built on.)"""
    import os
    torch.manual_seed(3)
    net = pkg.OthelloResNet(5, 64, board_size=6).eval()
    rng = np.random.Generator(np.random.PCG64(17))
    old = os.environ.get("OTH_WINO6")
    try:
        for n in (1, 7, 8, 9, 1000):
            occ = rng.random((n, 6, 6)) < 0.5
            own = occ & (rng.random((n, 6, 6)) < 0.5)
            x = torch.from_numpy(np.stack([own, occ & ~own, (~occ) & (rng.random((n, 6, 6)) < 0.4)], 1).astype(np.float32)).cuda()
            with torch.no_grad():
                rl, rv = net.cuda()(x)
            net.cpu()
            outs = {}
            for mode in ("0", "1"):
                os.environ["OTH_WINO6"] = mode
                ev = pkg.HipResNetEvaluator(net, precision="f16x3")
                logp, v = ev.forward_planes(x)
                torch.cuda.synchronize()
                assert (logp - rl).abs().max().item() < 1e-4 and (v - rv).abs().max().item() < 1e-4, (n, mode)
                outs[mode] = (logp.clone(), v.clone())
            assert (outs["0"][0] - outs["1"][0]).abs().max().item() < 1e-5, n
            assert ev.kernel_info(n)["kernel"].startswith("k_trunk_w6")
            if n > 1:   # the first position alone == the first position inside the batch (k_trunk_w6, mode "1")
                l1, v1 = ev.forward_planes(x[:1])
                torch.cuda.synchronize()
                assert torch.equal(l1[0], outs["1"][0][0]) and torch.equal(v1[0], outs["1"][1][0]), n
    finally:
        if old is None:
            os.environ.pop("OTH_WINO6", None)
        else:
            os.environ["OTH_WINO6"] = old


def _trained_like(pkg, blocks, filters, board, boost=1.0, seed=123):
    """Non-trivial BatchNorm statistics, uneven per-channel scales, peaked policies (what a trained checkpoint looks like);
    boost > 1: the stem's BatchNorm scaled up and the two head convolutions scaled down by the same factor, so the whole
    trunk's activations are ~boost times larger while the outputs stay those of an ordinary network."""
    torch.manual_seed(seed)
    net = pkg.OthelloResNet(blocks, filters, board_size=board).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 1.2 + 0.1)
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 1.8 + 0.3)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(torch.exp(torch.randn(mod.weight.shape[0], 1, 1, 1, generator=g) * 0.45))
        net.policy_head.fc.weight.mul_(2.0)
        if boost != 1.0:
            net.conv_block.bn.weight.mul_(boost)
            net.conv_block.bn.bias.mul_(boost)
            net.policy_head.conv.weight.div_(boost)
            net.value_head.conv.weight.div_(boost)
    return net


def _max_activation(net, x):
    """largest post-ReLU activation of the trunk under torch fp32 (what the fp16-split kernels clamp)"""
    mx = [0.0]
    hooks = [m.register_forward_hook(lambda _m, _i, o: mx.__setitem__(0, max(mx[0], float(o.max()))))
             for m in [net.conv_block] + list(net.res_blocks)]
    with torch.no_grad():
        out = net(x)
    for h in hooks:
        h.remove()
    return mx[0], out


@pytest.mark.parametrize("blocks,filters,board", [(6, 128, 8), (5, 64, 6), (3, 64, 8), (2, 32, 6)])
def test_saturated_launch_is_rescued_by_a_lower_activation_scale(pkg, blocks, filters, board):
    """The fp16-split trunks clamp activations at 1875 (Winograd trunks) / 3750 (direct ones) at their default activation
    scale 16; the reference's fp32 forward (net.py:182-205) has no clamp.  Trained-like weights scaled until the trunk's
    activations are in the thousands: the evaluator must NOT raise and must NOT return clamped outputs -- it halves the
    scale until the range fits (needs_rescue), says which scale it chose, and the result is within 1e-4 of torch fp32.
    An ordinary network never touches any of this."""
    import warnings
    n = 130
    x = (torch.rand(n, 3, board, board, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    plain = _trained_like(pkg, blocks, filters, board)
    ev0 = pkg.HipResNetEvaluator(plain)
    assert ev0.precision == "f16x3" and ev0.act_scale == 16.0
    ev0.forward_planes(x)
    assert not ev0.saturated() and ev0.rescues == [] and ev0.act_scale == 16.0
    # boost so that the largest activation is ~6000: beyond both clamps at scale 16, inside both at scale 4
    amax1, _ = _max_activation(plain.cuda(), x)
    plain.cpu()
    net = _trained_like(pkg, blocks, filters, board, boost=6000.0 / amax1)
    amax, (rl, rv) = _max_activation(net.cuda(), x)
    net.cpu()
    assert 3750 * 1.1 < amax < 7500 * 0.9, amax
    ev = pkg.HipResNetEvaluator(net)
    clamp16 = ev.kernel_info(n)["clamp"]
    assert clamp16 in (1875.0, 3750.0)
    # (1) the raw launch reports the clamp (rescue=False: one asynchronous launch, flag left for the caller) ...
    ev.forward_planes(x, rescue=False)
    with pytest.raises(pkg._lib.OthelloHipError):
        ev.check_saturation()
    assert not ev.saturated()                              # ... once: reading clears it
    # (2) the default call rescues: no exception, a lower scale, outputs within tolerance
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        logp, v = ev.forward_planes(x)
    info = ev.kernel_info(n)
    assert ev.precision == "f16x3" and ev.act_scale < 16.0 and info["act_scale"] == ev.act_scale
    assert ev.act_scale == 16.0 / 2 ** len(ev.rescues)
    assert info["clamp"] == clamp16 * 16.0 / ev.act_scale and info["clamp"] > amax
    assert len(ev.rescues) == len(wlist) >= 1 and "activation scale" in str(wlist[0].message)
    e1, e2 = (logp - rl).abs().max().item(), (v - rv).abs().max().item()
    print("\n    %dx%d on %dx%d, largest activation %.0f: rescued %d time(s) -> activation scale %g (clamp %g); max |dlogp| "
          "%.2e, max |dv| %.2e vs torch fp32" % (blocks, filters, board, board, amax, len(ev.rescues), ev.act_scale,
                                                 info["clamp"], e1, e2))
    assert e1 < 1e-4 and e2 < 1e-4
    assert not ev.saturated()
    # (3) the scale is sticky across weight reloads, and the same call is now quiet
    ev.refresh(force=True)
    k = len(ev.rescues)
    l2, v2 = ev.forward_planes(x)
    assert len(ev.rescues) == k and torch.equal(l2, logp) and torch.equal(v2, v)


def test_saturation_beyond_scale_one_falls_back_to_fp32(pkg):
    """Activations of ~1e6 are beyond the fp16-split kernels at ANY scale (30 000 / 60 000 at scale 1): the evaluator walks
    the scale down to 1 and then repacks for the exact-fp32 MFMA trunk -- still no exception."""
    import warnings
    torch.manual_seed(9)
    net = pkg.OthelloResNet(2, 64).eval()
    n = 130
    x = (torch.rand(n, 3, 8, 8, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    with torch.no_grad():
        net.conv_block.bn.weight.mul_(1.0e6)
    ev = pkg.HipResNetEvaluator(net)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        logp, v = ev.forward_planes(x)
    assert ev.precision == "f32" and len(ev.rescues) == 5 == len(wlist) and "exact-fp32" in ev.rescues[-1]
    assert ev.kernel_info(n)["kernel"].startswith("k_trunk_f32")
    with torch.no_grad():
        rl, rv = net.cuda()(x)
    net.cpu()
    # activations of ~1e6 here: fp32 rounding scales with them, so the check is relative (and the tanh is saturated)
    assert (logp - rl).abs().max().item() < 1e-4 * max(1.0, rl.abs().max().item()) and (v - rv).abs().max().item() < 2e-3


def _run_stream_steps(eng, steps, rescued):
    out = []
    for m in steps:
        g, ns = (eng.stream_step_rescued(m) if rescued else eng.stream_step(m))
        out.append((eng.game_ids().copy(),) + tuple(a.copy() for a in eng.selfplay_fetch(ns)[:3]))
    return out


def test_rescue_replays_the_call_from_its_start_state(pkg):
    """What 're-run the call from its start state' means for every kind of call, checked exactly: a worker whose
    evaluator has to be rescued in the middle of its work returns the SAME tuples, bit for bit, as one whose evaluator
    had the final activation scale from the start -- batch run (restart from the seed), stream step (snapshot / restore
    of the slots, the queued roots, the status words and the history ring's bookkeeping; twice, the second time with
    games in flight and finished games already harvested), lock-step search of the numpy-RNG worker, stand-alone search."""
    import warnings
    x = (torch.rand(64, 3, 8, 8, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    plain = _trained_like(pkg, 2, 64, 8)
    amax1, _ = _max_activation(plain.cuda(), x)
    net = _trained_like(pkg, 2, 64, 8, boost=6000.0 / amax1)   # k_trunk_h3: clamp 3750 at scale 16, 7500 at 8

    def fresh(scale=None):
        ev = pkg.HipResNetEvaluator(net)
        if scale is not None:
            pkg._lib.call("oth_net_set_act_scale", ev.handle, C.c_float(scale))
        return ev

    def set_scale(ev, s):
        pkg._lib.call("oth_net_set_act_scale", ev.handle, C.c_float(s))

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # ---- find the scale the rescue settles at (self-play visits positions the probe batch above does not)
        ev = fresh()
        eng = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev)
        n = eng.selfplay_run_rescued(24, seed=5)
        a = eng.selfplay_fetch(n)
        final = ev.act_scale
        assert final < 16.0 and len(ev.rescues) >= 1
        # batch run: identical to a run at the final scale from the start
        ev_ref = fresh(final)
        eng_ref = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev_ref)
        b = eng_ref.selfplay_fetch(eng_ref.selfplay_run(24, seed=5))
        assert ev_ref.rescues == [] and not ev_ref.saturated()
        assert all(np.array_equal(p, q) for p, q in zip(a, b))
        # ---- streaming: step 1 is rescued from the stream's start, step 3 (scale put back to 16 by hand) with games in
        #      flight, finished games harvested and the ring partly released
        steps = (10, 7, 9, 5)
        ref = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev_ref)
        ref.stream_begin(77, stagger_rounds=5)
        want = _run_stream_steps(ref, steps, rescued=False)
        ev2 = fresh()
        e2 = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev2)
        e2.stream_begin(77, stagger_rounds=5)
        got = _run_stream_steps(e2, steps[:2], rescued=True)
        assert ev2.act_scale == final and len(ev2.rescues) >= 1
        set_scale(ev2, 16.0)
        k = len(ev2.rescues)
        got += _run_stream_steps(e2, steps[2:], rescued=True)
        assert ev2.act_scale == final and len(ev2.rescues) > k
        for w_, g_ in zip(want, got):
            assert all(np.array_equal(p, q) for p, q in zip(w_, g_))
        assert e2.counters() == ref.counters()
        # ---- lock-step search (the numpy-RNG worker's path) and the stand-alone search.  (A search from the opening visits
        #      tamer positions than whole games: the scale it settles at may be higher than `final`; the reference runs at
        #      whatever scale the rescued call ended with.)
        ev3 = fresh()
        e3 = pkg.SearchEngine(4, 6, evaluator=ev3)
        e3.selfplay_begin(4)
        pi, act = e3.selfplay_search_rescued()
        s3 = ev3.act_scale
        assert s3 < 16.0 and len(ev3.rescues) >= 1
        ev_r3 = fresh(s3)
        r3 = pkg.SearchEngine(4, 6, evaluator=ev_r3)
        r3.selfplay_begin(4)
        pi_ref, act_ref = r3.selfplay_search()
        assert np.array_equal(pi, pi_ref) and np.array_equal(act, act_ref)
        set_scale(ev3, 16.0)
        actions = pi.argmax(1).astype(np.int32)
        e3.selfplay_apply(actions)
        r3.selfplay_apply(actions)
        k = len(ev3.rescues)
        pi, _ = e3.selfplay_search_rescued()             # rescued again, one ply into the games
        assert len(ev3.rescues) > k
        set_scale(ev_r3, ev3.act_scale)
        pi_ref, _ = r3.selfplay_search()
        assert np.array_equal(pi, pi_ref) and not ev_r3.saturated()
        roots = game_positions(3, 5)[:8]
        s, o = [p[0] for p in roots], [p[1] for p in roots]
        set_scale(ev3, 16.0)
        k = len(ev3.rescues)
        e4 = pkg.SearchEngine(8, 10, evaluator=ev3)
        e4.search_begin(s, o)
        e4.search_run_rescued()
        assert len(ev3.rescues) > k
        set_scale(ev_r3, ev3.act_scale)
        r4 = pkg.SearchEngine(8, 10, evaluator=ev_r3)
        r4.search_begin(s, o)
        r4.search_run()
        assert not ev_r3.saturated()
        assert all(np.array_equal(p, q) for p, q in zip(e4.search_results(1.0), r4.search_results(1.0)))
        # ---- the drop-in worker, end to end: no exception, same tuples as a worker that never needed a rescue
        np.random.seed(11)
        w1 = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=4, num_parallel_games=8, verbose=False)
        d1 = w1.execute_episodes(6)
        assert w1.batch_mcts.evaluator.act_scale < 16.0 and len(w1.batch_mcts.evaluator.rescues) >= 1
        np.random.seed(11)
        w2 = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=4, num_parallel_games=8, verbose=False)
        set_scale(w2.batch_mcts.evaluator, w1.batch_mcts.evaluator.act_scale)
        d2 = w2.execute_episodes(6)
        assert w2.batch_mcts.evaluator.rescues == []
        assert len(d1) == len(d2) and all(np.array_equal(p[0], q[0]) and np.array_equal(p[1], q[1]) and p[2] == q[2]
                                          for p, q in zip(d1, d2))


def test_weight_update_resets_the_activation_scale(pkg):
    """The activation scale is a function of the WEIGHTS, not of the process's history (it used to be sticky -- one
    saturating checkpoint lowered it, or switched the evaluator to fp32, for good).  When the trainer has stepped (the version
    counters moved), refresh() goes back to scale 16 and the constructor's precision, uploads, and one launch over 256 fixed
    probe positions lets the rescue lower the scale again if the NEW weights need it."""
    import warnings
    x = (torch.rand(64, 3, 8, 8, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    tame = _trained_like(pkg, 2, 64, 8)
    amax1, _ = _max_activation(tame.cuda(), x)
    tame.cpu()
    wild = _trained_like(pkg, 2, 64, 8, boost=6000.0 / amax1)
    model = pkg.OthelloResNet(2, 64).eval()
    model.load_state_dict(wild.state_dict())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, model, num_simulations=4, num_parallel_games=8, verbose=False)
        ev = w.batch_mcts.evaluator
        np.random.seed(11)
        w.execute_episodes(6)
        low = ev.act_scale
        assert low < 16.0 and len(ev.rescues) >= 1 and w.last_stats["act_scale"] == low and w.last_stats["rescues"] == len(ev.rescues)
        # the trainer steps to tame weights: the scale goes back up, nothing is rescued
        with torch.no_grad():
            for p_, q_ in zip(list(model.parameters()) + list(model.buffers()), list(tame.parameters()) + list(tame.buffers())):
                p_.copy_(q_)
        k = len(ev.rescues)
        np.random.seed(11)
        d_tame = w.execute_episodes(6)
        assert ev.act_scale == 16.0 and ev.precision == "f16x3" and len(ev.rescues) == k
        # ... identical to a worker that never saw the wild weights
        np.random.seed(11)
        w2 = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, tame, num_simulations=4, num_parallel_games=8, verbose=False)
        d_ref = w2.execute_episodes(6)
        assert len(d_tame) == len(d_ref) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
                                                 for a, b in zip(d_tame, d_ref))
        # ... and back to the wild ones: the probe launch of refresh() settles the scale BEFORE the call (no replayed call)
        with torch.no_grad():
            for p_, q_ in zip(list(model.parameters()) + list(model.buffers()), list(wild.parameters()) + list(wild.buffers())):
                p_.copy_(q_)
        ev.refresh()
        assert ev.act_scale < 16.0 and len(ev.rescues) > k
        # a forced re-upload of unchanged weights keeps the scale (no version change)
        s_now = ev.act_scale
        ev.refresh(force=True)
        assert ev.act_scale == s_now


def test_rescue_replay_with_the_evaluation_cache(pkg):
    """The rescue's snapshot / restore with the evaluation cache ON (restore also clears the cache, resets its
    epoch and puts back the cres / cstat rows, and that path was never replay-tested).  A stream whose evaluator is rescued at
    its start and again in mid-stream (scale put back to 16 by hand, games in flight) returns the tuples, the counters AND the
    cache statistics of a stream that had the final scale from the start; the lock-step search likewise (a replayed search
    starts from an empty cache -- the abandoned call's entries may hold clamped rows -- so the reference search is started from
    an empty cache too, by the same snapshot / restore pair)."""
    import warnings
    x = (torch.rand(64, 3, 8, 8, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    plain = _trained_like(pkg, 2, 64, 8)
    amax1, _ = _max_activation(plain.cuda(), x)
    net = _trained_like(pkg, 2, 64, 8, boost=6000.0 / amax1)

    def set_scale(ev, s_):
        pkg._lib.call("oth_net_set_act_scale", ev.handle, C.c_float(s_))

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev0 = pkg.HipResNetEvaluator(net)
        e0 = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev0)
        e0.selfplay_run_rescued(24, seed=5)
        final = ev0.act_scale
        assert final < 16.0
        # ---- streaming, cache of 2^12 entries
        steps = (10, 7, 9, 5)
        ev_ref = pkg.HipResNetEvaluator(net)
        set_scale(ev_ref, final)
        ref = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev_ref, eval_cache_log2=12)
        ref.stream_begin(77, stagger_rounds=5)
        want = _run_stream_steps(ref, steps, rescued=False)
        assert ev_ref.rescues == [] and ref.counters()["cache_hits"] > 0
        ev2 = pkg.HipResNetEvaluator(net)
        e2 = pkg.SearchEngine(16, 6, temperature_threshold=8, evaluator=ev2, eval_cache_log2=12)
        e2.stream_begin(77, stagger_rounds=5)
        got = _run_stream_steps(e2, steps[:2], rescued=True)
        assert ev2.act_scale == final and len(ev2.rescues) >= 1
        set_scale(ev2, 16.0)
        k = len(ev2.rescues)
        got += _run_stream_steps(e2, steps[2:], rescued=True)
        assert ev2.act_scale == final and len(ev2.rescues) > k
        for w_, g_ in zip(want, got):
            assert all(np.array_equal(p, q) for p, q in zip(w_, g_))
        assert e2.counters() == ref.counters() and e2.cache_stats() == ref.cache_stats()
        # ---- lock-step (the numpy-RNG worker's path), cache on: search, apply, search again with a rescue in between
        ev3 = pkg.HipResNetEvaluator(net)
        e3 = pkg.SearchEngine(4, 6, evaluator=ev3, eval_cache_log2=10)
        e3.selfplay_begin(4)
        pi1, act1 = e3.selfplay_search_rescued()
        s3 = ev3.act_scale
        ev_r3 = pkg.HipResNetEvaluator(net)
        set_scale(ev_r3, s3)
        r3 = pkg.SearchEngine(4, 6, evaluator=ev_r3, eval_cache_log2=10)
        r3.selfplay_begin(4)
        pr1, ar1 = r3.selfplay_search()
        assert np.array_equal(pi1, pr1) and np.array_equal(act1, ar1) and e3.cache_stats() == r3.cache_stats()
        actions = pi1.argmax(1).astype(np.int32)
        e3.selfplay_apply(actions)
        r3.selfplay_apply(actions)
        set_scale(ev3, 16.0)
        k = len(ev3.rescues)
        pi2, _ = e3.selfplay_search_rescued()
        assert len(ev3.rescues) > k
        set_scale(ev_r3, ev3.act_scale)
        r3.snapshot()
        r3.restore()                                     # the replay's start state: this search's roots, an empty cache
        pr2, _ = r3.selfplay_search()
        assert np.array_equal(pi2, pr2) and not ev_r3.saturated()
        assert e3.counters() == r3.counters() and e3.cache_stats() == r3.cache_stats()


def test_bench_workload_rescues_across_lanes(pkg):
    """bench.py's own step (Workload.play): two lanes on two streams and host threads share ONE evaluator, every lane
    snapshots its stream before the step, and the rescue of a saturated launch is decided where both lanes have joined --
    both are then restored and the step is replayed.  With a network that saturates at the default activation scale the
    steps must return, lane by lane, the tuples of a workload whose evaluator had the settled scale from the start."""
    import importlib.util
    import os
    import types
    import warnings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    x = (torch.rand(64, 3, 8, 8, generator=torch.Generator().manual_seed(2)) < 0.35).float().cuda()
    plain = _trained_like(pkg, 2, 64, 8)
    amax1, _ = _max_activation(plain.cuda(), x)
    net = _trained_like(pkg, 2, 64, 8, boost=6000.0 / amax1)
    proxy = types.SimpleNamespace(**{k: getattr(pkg, k) for k in dir(pkg) if not k.startswith("__")})
    proxy.OthelloResNet = lambda *a_, **k_: net           # Workload builds its network by seed: hand it the boosted one

    def set_scale(ev, s_):
        pkg._lib.call("oth_net_set_act_scale", ev.handle, C.c_float(s_))

    def tuples(parts):
        return [tuple(t_.clone() for t_ in lane) for lane in parts]

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = bench.Workload(proxy, torch, 8, 2, 64, 6, 32, 2, 5, 16)      # 32 slots in two lanes, 6 sims, steps of 16 games
        g1, p1 = w.play(16)
        t1 = tuples(p1)
        s1 = w.ev.act_scale
        assert s1 < 16.0 and len(w.ev.rescues) >= 1 and g1 >= 16
        set_scale(w.ev, 16.0)                                                # the second step saturates again, in mid-stream
        k = len(w.ev.rescues)
        g2, p2 = w.play(16)
        t2 = tuples(p2)
        s2 = w.ev.act_scale
        assert len(w.ev.rescues) > k and s2 < 16.0
        ref = bench.Workload(proxy, torch, 8, 2, 64, 6, 32, 2, 5, 16)
        set_scale(ref.ev, s1)
        h1, q1 = ref.play(16)
        u1 = tuples(q1)
        set_scale(ref.ev, s2)
        h2, q2 = ref.play(16)
        u2 = tuples(q2)
        assert ref.ev.rescues == [] and (g1, g2) == (h1, h2)
    for got, want in ((t1, u1), (t2, u2)):
        assert len(got) == len(want) == 2
        for lane_g, lane_w in zip(got, want):
            assert all(torch.equal(a_, b_) for a_, b_ in zip(lane_g, lane_w))
    assert w.counters() == ref.counters()


def test_lane_streams_are_made_once(pkg):
    """The lanes' streams are a per-process pool (engine.lane_streams), not new streams per run: with new ones the two streams
    of every second two-lane run shared a hardware queue and the run lost 8-10 % (profiles/r05_lane_modes.log).  The multi-lane
    worker and bench.py's Workload both draw from the pool."""
    from othello_reinforcement_learning_test_amd.engine import lane_streams
    a = lane_streams(2)
    b = lane_streams(3)
    assert a[0] is b[0] and a[1] is b[1] and len(b) == 3 and len({s.cuda_stream for s in b}) == 3
    assert lane_streams(2)[1] is a[1]
    torch.manual_seed(3)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=4, num_parallel_games=16, lanes=2, verbose=False)
    np.random.seed(1)
    d1 = w.execute_episodes(12)
    np.random.seed(1)
    d2 = w.execute_episodes(12)                      # second call: the same two streams, the same tuples
    assert len(d1) == len(d2) > 0 and all(np.array_equal(x[1], y[1]) and x[2] == y[2] for x, y in zip(d1, d2))
    assert lane_streams(2)[0] is a[0]


def test_lane_overlap_is_a_checked_property(pkg):
    """Two lanes whose streams share a hardware queue serialise (round 5: -8...-10 %), and nothing used to notice.  bench.py's
    Workload (and the multi-lane worker) now MEASURE the overlap on the first warm-up step -- sum of the trunk launch durations
    / union of their intervals, from the HIP-event spans: the normal two-lane run reports >= 1.5; with both lanes bound to ONE
    stream the detector flags it (below the floor of 1.3: RuntimeWarning, `lanes_serialised`), draws the streams once more from a widened pool
    and the next step overlaps again.  (10x128 network, 2 x 2048 slots as the headline: launches of ~2 ms -- measured 1.83 on
    separate queues; at 2 x 1024 slots the figures were 1.51 / 1.18 / 1.67, too close to the bars; toy networks are launch-bound
    and are never flagged.)"""
    import importlib.util
    import os
    import warnings
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    E = pkg.engine

    def decide(w_):
        """steps with the check on until it has decided on the current arrangement (>= 100 launches per lane seen)"""
        k, n = len(w_.check.history), 0
        while len(w_.check.history) == k:
            w_.play(64, check=True)
            n += 1
            assert n < 80
        return w_.check.history[-1][1]

    w = bench.Workload(pkg, torch, 8, 10, 128, 10, 4096, 2, 8, 64)
    m = decide(w)
    rep = w.check.report()
    print("\n    two lanes on their own streams: overlap %.2f (%d launches, mean %.2f ms)" % (m["overlap"], m["launches"], m["mean_launch_ms"]))
    assert rep["lanes_overlap"] >= 1.5 and rep["lanes_serialised"] is False and rep["stream_redraws"] == 0
    assert m["launches"] >= 2 * E.OVERLAP_MIN_LAUNCHES
    assert not w.check.pending and not any(e.timing for e in w.engs)       # decided once; the hooks are off again
    t0 = [tuple(t_.clone() for t_ in lane) for lane in w.play(64)[1]]      # (an unchecked step: no hooks)
    # ---- both lanes on ONE stream, no redraw allowed: flagged
    torch.cuda.synchronize()
    one = w.streams[0]
    w.streams = [one, one]
    w.check = E.LaneOverlapCheck(2, w.dev, max_redraws=0)
    E._WARNED.clear()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        m = decide(w)
    rep = w.check.report()
    print("    both lanes on one stream:        overlap %.2f (%d launches, mean %.2f ms)" % (m["overlap"], m["launches"], m["mean_launch_ms"]))
    # (two lanes on ONE stream measure 1.13-1.20, not 1.0: the HIP-event intervals of consecutive launches of a stream overlap a
    # little; two streams on one hardware queue measure 1.00-1.02, profiles/r06_lane_redraw_ab*.log; the floor for two lanes is 1.3)
    assert rep["lanes_overlap"] < E.overlap_floor(2) and rep["lanes_serialised"] is True and not w.check.pending
    assert any("do not overlap" in str(x.message) for x in wl)
    # ---- the same, one redraw allowed: the streams are drawn again and the NEXT steps are measured on the new arrangement
    torch.cuda.synchronize()
    w.streams = [one, one]
    w.check = E.LaneOverlapCheck(2, w.dev, max_redraws=1)
    E._WARNED.clear()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        decide(w)
    assert any("do not overlap" in str(x.message) for x in wl), w.check.history
    assert w.check.pending and w.streams[0] is not w.streams[1] and one not in w.streams
    decide(w)
    rep = w.check.report()
    print("    after the redraw:                overlap %.2f (arrangements tried: %s)" % (rep["lanes_overlap"], rep["arrangements_tried"]))
    assert rep["stream_redraws"] == 1 and len(rep["arrangements_tried"]) == 2 and rep["arrangements_tried"][0] < E.overlap_floor(2)
    assert rep["lanes_overlap"] >= 1.5 and rep["lanes_serialised"] is False and not w.check.pending
    assert len(t0) == 2                                                    # (tuples do not depend on the streams: exact tests)
    w.close()


def test_trunk_on_trained_like_weights(pkg):
    """fp16x3 trunk with non-trivial BatchNorm statistics, uneven per-channel scales and peaked policies (what a
    trained checkpoint looks like, unlike the seeded-random init): within 1e-4 of torch fp32 and no noisier
    against a float64 forward than fp32 arithmetic itself; the single f16 pass is far outside the tolerance."""
    torch.manual_seed(123)
    net = pkg.OthelloResNet(6, 128).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for mod in net.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 1.2 + 0.1)
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 1.8 + 0.3)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
            if isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(torch.exp(torch.randn(mod.weight.shape[0], 1, 1, 1, generator=g) * 0.45))
        net.policy_head.fc.weight.mul_(2.0)
    pos = game_positions(8, 31)[:400]
    s = np.array([p[0] for p in pos], dtype=U64)
    o = np.array([p[1] for p in pos], dtype=U64)
    ds, do = dev_u64(s), dev_u64(o)
    lg = pkg.DeviceBoards.legal_moves(ds, do)
    x = pkg.DeviceBoards.tensor_input(ds, do)
    with torch.no_grad():
        rl, rv = net.cuda()(x)
        net64 = pkg.OthelloResNet(6, 128).eval().double()
        net64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
        tl, tv = net64.cuda()(x.double())
    net.cpu()
    assert rl.exp().max().item() > 0.3 and tl.min().item() > -60   # peaked but sane policies
    noise = (rl.double() - tl).abs().max().item()                  # fp32 arithmetic's own distance from float64
    errs = {}
    for prec in ("f16x3", "f16x3_direct", "f32", "f16"):
        ev = pkg.HipResNetEvaluator(net, precision=prec)
        logp, v = ev.forward_bits(ds, do, lg)
        errs[prec] = ((logp - rl).abs().max().item(), (v - rv).abs().max().item(),
                      (logp.double() - tl).abs().max().item())
    print("trained-like net: torch fp32 vs float64 %.2e; max |dlogp| vs torch fp32: f16x3 (Winograd) %.2e, f16x3_direct %.2e, "
          "f32 %.2e, single f16 %.2e" % (noise, errs["f16x3"][0], errs["f16x3_direct"][0], errs["f32"][0], errs["f16"][0]))
    for prec in ("f16x3", "f16x3_direct", "f32"):
        assert errs[prec][0] < 1e-4 and errs[prec][1] < 1e-4, (prec, errs[prec])
        assert errs[prec][2] < 3 * noise + 1e-6, (prec, errs[prec], noise)
    assert errs["f16"][0] > 1e-4     # why the single pass is not the default


def test_net_weight_refresh(pkg):
    """The trainer mutates the model between calls; the evaluator must pick the new weights up."""
    torch.manual_seed(1)
    net = pkg.OthelloResNet(2, 16).eval()
    ev = pkg.HipResNetEvaluator(net)
    x = torch.from_numpy(np.stack([ol.tensor(ol.board())])).cuda()
    a, _ = ev.forward_planes(x)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ev.refresh()
    b, _ = ev.forward_planes(x)
    with torch.no_grad():
        ref, _ = net(x.cpu())
    assert not torch.equal(a, b) and (b.cpu() - ref).abs().max().item() < 1e-4


# =========================================================================================== self-play
# (the episode-stream parity tests live in tests/test_gpu_selfplay_exact.py: exact, oracle driven by the HIP network)


def _check_replay_consistency(pkg, st, pi, z, game_len, threshold, onehot_late):
    """Domain properties of a device-RNG run, checked with the oracle's rules: every game starts at
    the initial position, each recorded state is reached from the previous one by a legal move that
    had pi > 0, plane 2 is the legal mask, pi sums to 1 on legal moves, z = winner*player (L16)."""
    off = 0
    for glen in game_len:
        b = ol.board()
        for p in range(glen):
            s = st[off + p]
            assert np.array_equal(s, ol.tensor(b)), "state %d of game is not the oracle's position" % p
            legal = ol.legal_list(b)
            assert abs(pi[off + p].sum() - 1.0) < 1e-5
            assert all(pi[off + p][a] == 0 for a in range(65) if a not in legal)
            if onehot_late and p >= threshold:
                assert (pi[off + p] != 0).sum() == 1
            # find the move played: the legal action with pi>0 leading to the next recorded state
            if p + 1 < glen:
                nxt = st[off + p + 1]
                found = False
                for a in legal:
                    if pi[off + p][a] <= 0:
                        continue
                    c = ol.board(b.self_board, b.opp_board, b.move_count)
                    ol.lib().orc_make_move(c, a)
                    if np.array_equal(ol.tensor(c), nxt):
                        b, found = c, True
                        break
                assert found, "no legal move explains the next state"
            else:
                ends = []
                for a in legal:
                    if pi[off + p][a] <= 0:
                        continue
                    c = ol.board(b.self_board, b.opp_board, b.move_count)
                    ol.lib().orc_make_move(c, a)
                    if ol.lib().orc_is_terminal(c):
                        ends.append(ol.lib().orc_winner(c))
                assert ends, "last recorded ply does not end the game"
                zs = {tuple(float(w * (1 if q % 2 == 0 else -1)) for q in range(glen)) for w in ends}
                assert tuple(z[off:off + glen].tolist()) in zs
        off += glen
    assert off == len(z)


def test_selfplay_device_rng_properties(pkg):
    torch.manual_seed(3)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=6, temperature_threshold=8,
                                   num_parallel_games=16, verbose=False, device_slots=16)
    np.random.seed(0)
    data = w.execute_episodes(40)            # 40 games through 16 slots: refill path
    assert isinstance(data, list) and isinstance(data[0], tuple)
    s0, p0, z0 = data[0]
    assert s0.shape == (3, 8, 8) and s0.dtype == np.float32 and p0.shape == (65,) and isinstance(z0, float)
    assert z0 in (-1.0, 0.0, 1.0)            # reference tests/test_train.py:116-130
    eng = w.engine
    st, pi, z, gl = eng.selfplay_fetch(len(data))
    assert len(gl) == 40 and gl.sum() == len(data) and gl.min() >= 9
    _check_replay_consistency(pkg, st, pi, z, gl, 8, onehot_late=False)
    c = w.last_stats
    assert c["games"] == 40 and c["plies"] == len(data) and c["simulations"] == 6 * len(data)
    assert c["evals"] == len(data) + c["simulations"] - c["terminal_sims"]
    # same numpy seed => same run; different seed => different games
    np.random.seed(0)
    again = w.execute_episodes(40)
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
               for a, b in zip(data, again))
    other = w.execute_episodes(40)
    assert len(other) != len(data) or any(not np.array_equal(a[0], b[0]) for a, b in zip(data, other))
    # the slot count is a speed knob only: the default (engine grown to the call's 40 episodes -> 64 slots, no refill)
    # returns the same tuples as the 16-slot engine for the same seed
    w_auto = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=6, temperature_threshold=8,
                                        num_parallel_games=16, verbose=False)
    np.random.seed(0)
    wide = w_auto.execute_episodes(40)
    assert w_auto.engine.max_games == 64 and len(wide) == len(data)
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(data, wide))


def test_selfplay_400_sims_deep_trees(pkg):
    """BASELINE.json configs[3] shape (400 sims/move, c_puct 1.5, threshold 20) on a small net and few
    games: exercises the large per-game arenas end to end; replay tuples must be consistent."""
    torch.manual_seed(9)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=400, temperature_threshold=20,
                                   num_parallel_games=8, c_puct=1.5, verbose=False)
    np.random.seed(1)
    data = w.execute_episodes(8)
    st, pi, z, gl = w.engine.selfplay_fetch(len(data))
    _check_replay_consistency(pkg, st, pi, z, gl, 20, onehot_late=False)
    c = w.last_stats
    assert c["simulations"] == 400 * len(data) and c["games"] == 8


def test_eval_cache_is_bit_identical(pkg):
    """Opt-in evaluation cache: same seed => identical replay tuples with the cache on and off (the network
    is a pure function of the position), with fewer network evaluations."""
    torch.manual_seed(11)
    net = pkg.OthelloResNet(2, 16).eval()
    outs, stats = [], []
    for lg in (0, 16):
        w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=20, temperature_threshold=8,
                                       num_parallel_games=32, verbose=False, eval_cache_log2=lg)
        np.random.seed(5)
        outs.append(w.execute_episodes(48))
        stats.append(dict(w.last_stats))
    assert len(outs[0]) == len(outs[1])
    for a, b in zip(*outs):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert stats[0]["cache_hits"] == 0 and stats[1]["cache_hits"] > 0
    assert stats[1]["evals"] + stats[1]["cache_hits"] == stats[0]["evals"]
    assert stats[1]["evals"] < 0.85 * stats[0]["evals"]
    # the statistics of the cache (round 5): every network evaluation is either the first one of its position since the
    # cache was cleared or a repeat; a table far too small for the run (2^10 entries) evicts live entries and re-evaluates
    # positions it had seen -- and the tuples are STILL identical
    cs = w.engine.cache_stats()
    assert cs["entries"] == 1 << 16 and cs["distinct_positions"] + cs["repeated_evals"] == stats[1]["evals"]
    assert cs["distinct_positions"] > 0 and cs["repeated_evals"] < 0.25 * stats[1]["evals"]
    w2 = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=20, temperature_threshold=8,
                                    num_parallel_games=32, verbose=False, eval_cache_log2=10)
    np.random.seed(5)
    small = w2.execute_episodes(48)
    assert all(np.array_equal(a_[0], b_[0]) and np.array_equal(a_[1], b_[1]) and a_[2] == b_[2] for a_, b_ in zip(outs[0], small))
    cs2 = w2.engine.cache_stats()
    assert cs2["entries"] == 1024 and cs2["conflict_evictions"] > 0 and cs2["repeated_evals"] > cs["repeated_evals"]
    assert cs2["distinct_positions"] + cs2["repeated_evals"] == w2.last_stats["evals"] > stats[1]["evals"]


def test_serial_worker_device_mode_onehot(pkg):
    torch.manual_seed(4)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.SelfPlayWorker(pkg.OthelloBitboard, pkg.MCTS(net), num_simulations=5, temperature_threshold=10)
    data = w.execute_episodes(6)
    eng = w._engine
    st, pi, z, gl = eng.selfplay_fetch(len(data))
    _check_replay_consistency(pkg, st, pi, z, gl, 10, onehot_late=True)


def test_trainer_contract_with_replay_buffer_shapes(pkg):
    """What trainer.py:180-185 + buffer.py:81-83 do with the result: np.array over the tuples."""
    torch.manual_seed(5)
    net = pkg.OthelloResNet(2, 16).eval()
    cfg = {"mcts": {"num_simulations": 5, "c_puct": 1.0}, "self_play": {"temperature_threshold": 10,
                                                                        "num_parallel_games": 4}}
    w = pkg.create_parallel_self_play_worker(cfg, net, torch.device("cuda"), verbose=False)
    data = w.execute_episodes(num_episodes=3, add_dirichlet_noise=True)
    states = np.array([d[0] for d in data], dtype=np.float32)
    pols = np.array([d[1] for d in data], dtype=np.float32)
    vals = np.array([d[2] for d in data], dtype=np.float32).reshape(-1, 1)
    assert states.shape[1:] == (3, 8, 8) and pols.shape[1:] == (65,) and vals.shape[1:] == (1,)


# =========================================================================================== next rows
def test_augment_symmetries_device(pkg):
    """oth_augment_symmetries == OthelloBitboard.get_symmetries (literal transform incl. the legal plane)."""
    from othello_reinforcement_learning_test_amd.replay import augment_symmetries
    pos = game_positions(3, 21)[:97]
    rng = np.random.Generator(np.random.PCG64(2))
    st = np.stack([ol.tensor(ol.board(s, o)) for s, o, _ in pos])
    pi = rng.random((len(pos), 65)).astype(np.float32)
    z = rng.integers(-1, 2, len(pos)).astype(np.float32)
    so, po, zo = augment_symmetries(torch.from_numpy(st).cuda(), torch.from_numpy(pi).cuda(), torch.from_numpy(z).cuda())
    so, po, zo = so.cpu().numpy(), po.cpu().numpy(), zo.cpu().numpy()
    assert so.shape == (8 * len(pos), 3, 8, 8) and po.shape == (8 * len(pos), 65)
    for i, (s, o, _) in enumerate(pos):
        es, ep = ol.symmetries(ol.board(s, o), pi[i])
        assert np.array_equal(so[8 * i:8 * i + 8], es) and np.array_equal(po[8 * i:8 * i + 8], ep)
        assert np.all(zo[8 * i:8 * i + 8] == z[i])


def test_device_replay_buffer_from_engine(pkg):
    """Self-play tuples go from the engine into the device ring without touching the host."""
    from othello_reinforcement_learning_test_amd.replay import DeviceReplayBuffer
    torch.manual_seed(6)
    net = pkg.OthelloResNet(2, 16).eval()
    eng = pkg.SearchEngine(16, 5, temperature_threshold=8, evaluator=pkg.HipResNetEvaluator(net))
    n = eng.selfplay_run(16, seed=3)
    st, pi, z = eng.selfplay_device_tensors()
    assert st.is_cuda and st.shape[0] == n
    buf = DeviceReplayBuffer(max_size=500, device="cuda")
    buf.add((st, pi, z))
    assert len(buf) == min(500, n)
    s, p, v = buf.sample(64)
    assert s.is_cuda and tuple(s.shape) == (64, 3, 8, 8) and tuple(v.shape) == (64, 1)
    hs, hp, hz, _ = eng.selfplay_fetch(n)
    o_s, o_p, o_z = buf.ordered()
    k = len(buf)
    assert np.array_equal(o_s.cpu().numpy(), hs[n - k:]) and np.array_equal(o_z.cpu().numpy(), hz[n - k:])
    # the minibatch gather kernel (oth_replay_gather) == indexing the ring, also after the ring has wrapped
    buf.add((st[:300], pi[:300], z[:300]))
    o_s, o_p, o_z = buf.ordered()
    torch.manual_seed(9)
    s, p, v = buf.sample(128)
    torch.manual_seed(9)
    pick = torch.randperm(len(buf), device="cuda")[:128]
    assert torch.equal(s, o_s[pick]) and torch.equal(p, o_p[pick]) and torch.equal(v[:, 0], o_z[pick])
    s, p, v = buf.sample(len(buf))      # without replacement: a permutation of the whole ring
    assert torch.equal(torch.sort(v[:, 0]).values, torch.sort(o_z).values)
    assert abs(float(p.sum()) - float(o_p.sum())) < 1e-2
    # trainer-side feed: a few optimisation steps straight from the device ring (trainer.py:243-328)
    from othello_reinforcement_learning_test_amd.replay import train_epochs
    model = pkg.OthelloResNet(2, 16).cuda()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    l0 = train_epochs(model, opt, buf, num_epochs=2, batch_size=64)
    l1 = train_epochs(model, opt, buf, num_epochs=20, batch_size=64)
    assert np.isfinite(l0) and l1 < l0


def test_batched_arena_equals_sequential(pkg):
    """Lock-step batched evaluation (one device search per ply over all boards) gives, against a deterministic
    opponent, exactly the matches the reference-style sequential Arena plays with MCTSPlayer."""
    from othello_reinforcement_learning_test_amd.arena import Arena, BatchedArena, GreedyPlayer, MCTSPlayer
    torch.manual_seed(8)
    net = pkg.OthelloResNet(2, 16).eval()
    mp = MCTSPlayer(net, torch.device("cuda"), num_simulations=12, name="AI")
    seq = Arena(verbose=False).play_matches(mp, GreedyPlayer("G"), num_games=6, alternate_colors=True)
    bm = pkg.BatchMCTS(net, c_puct=1.0, evaluator=mp.mcts.evaluator)
    bat = BatchedArena(bm, num_simulations=12).play_matches("AI", GreedyPlayer("G"), num_games=6)
    for a, b in zip(seq, bat):
        assert (a.winner, a.player1_score, a.player2_score, a.num_moves) == \
               (b.winner, b.player1_score, b.player2_score, b.num_moves)
    assert all(r.num_moves >= 9 for r in seq)


def test_training_iterations_end_to_end(pkg):
    """The loop of trainer.py:165-226 around the engine: self-play -> replay buffer -> SGD steps on the torch
    module (policy cross-entropy + value MSE as trainer.py:330-364) -> the next self-play call must see the new
    weights.  Two iterations with the test.yaml-sized network."""
    from othello_reinforcement_learning_test_amd.replay import DeviceReplayBuffer
    torch.manual_seed(0)
    np.random.seed(0)
    net = pkg.OthelloResNet(2, 16).cuda()
    cfg = {"mcts": {"num_simulations": 5}, "self_play": {"temperature_threshold": 10, "num_parallel_games": 8}}
    worker = pkg.create_parallel_self_play_worker(cfg, net, torch.device("cuda"), verbose=False)
    buf = DeviceReplayBuffer(max_size=2000, device="cuda")
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    first_version = None
    for it in range(2):
        net.eval()
        data = worker.execute_episodes(num_episodes=8, add_dirichlet_noise=True)
        buf.add(data)
        assert buf.is_ready(16)
        net.train()
        for _ in range(3):
            s, p, v = buf.sample(16)
            logp, val = net(s)
            loss = -(p * logp).sum(1).mean() + torch.nn.functional.mse_loss(val, v)
            opt.zero_grad()
            loss.backward()
            opt.step()
        assert torch.isfinite(loss)
        ver = worker.batch_mcts.evaluator._version
        assert first_version is None or ver != first_version   # the evaluator re-read the stepped weights
        first_version = ver
    # after training the HIP forward still matches the (updated) torch module
    net.eval()
    x = torch.from_numpy(np.stack([ol.tensor(ol.board())])).cuda()
    worker.batch_mcts.evaluator.refresh()
    hl, hv = worker.batch_mcts.evaluator.forward_planes(x)
    with torch.no_grad():
        tl, tv = net(x)
    assert (hl - tl).abs().max().item() < 1e-4 and (hv - tv).abs().max().item() < 1e-4


def test_empty_and_tiny_inputs(pkg):
    """Edge cases: n = 0 batches are no-ops, one-slot engines and runs smaller than the slot count work."""
    DB = pkg.DeviceBoards
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    assert DB.legal_moves(e, e).numel() == 0 and DB.tensor_input(e, e).shape == (0, 3, 8, 8)
    ok, fl = DB.make_move(e.clone(), e.clone(), torch.empty(0, dtype=torch.int32, device="cuda"))
    assert ok.numel() == 0 and fl.numel() == 0
    torch.manual_seed(2)
    net = pkg.OthelloResNet(2, 16).eval()
    ev = pkg.HipResNetEvaluator(net)
    l, v = ev.forward_bits(e, e, e)
    assert l.shape == (0, 65) and v.shape == (0, 1)
    eng1 = pkg.SearchEngine(1, 3, temperature_threshold=4, evaluator=ev)          # a single slot
    n = eng1.selfplay_run(3, seed=9)                                              # 3 games through 1 slot
    st, pi, z, gl = eng1.selfplay_fetch(n)
    assert len(gl) == 3 and gl.sum() == n and n >= 27
    _check_replay_consistency(pkg, st, pi, z, gl, 4, onehot_late=False)
    eng = pkg.SearchEngine(64, 3, temperature_threshold=4, evaluator=ev)
    n = eng.selfplay_run(5, seed=9)                                               # fewer games than slots
    assert eng.selfplay_fetch(n)[3].shape == (5,)
    with pytest.raises(pkg.OthelloHipError):
        pkg.SearchEngine(4, 5000)                                                 # beyond the simulation cap
    from othello_reinforcement_learning_test_amd.replay import augment_symmetries
    so, po, zo = augment_symmetries(torch.empty(0, 3, 8, 8, device="cuda"), torch.empty(0, 65, device="cuda"),
                                    torch.empty(0, device="cuda"))
    assert so.shape == (0, 3, 8, 8) and zo.numel() == 0


def test_lanes_overlap_mode(pkg):
    """lanes=2: two engines on two streams/threads; tuples stay consistent and the counters add up."""
    torch.manual_seed(12)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=6, temperature_threshold=8,
                                   num_parallel_games=16, verbose=False, lanes=2)
    np.random.seed(3)
    data = w.execute_episodes(30)
    st = np.stack([d[0] for d in data]); pi = np.stack([d[1] for d in data])
    z = np.array([d[2] for d in data], dtype=np.float32)
    # recover the game lengths from the initial-position markers (every game starts from the start position)
    start = ol.tensor(ol.board())
    firsts = [i for i in range(len(st)) if np.array_equal(st[i], start) and (i == 0 or True)]
    c = w.last_stats
    assert c["games"] == 30 and c["plies"] == len(data) and c["simulations"] == 6 * len(data)
    gl = []
    i = 0
    while i < len(st):   # walk game by game with the oracle
        b = ol.board(); n = 0
        while i + n < len(st) and np.array_equal(st[i + n], ol.tensor(b)):
            nxt = None
            if i + n + 1 < len(st):
                for a in ol.legal_list(b):
                    if pi[i + n][a] > 0:
                        cb = ol.board(b.self_board, b.opp_board, b.move_count)
                        ol.lib().orc_make_move(cb, a)
                        if np.array_equal(ol.tensor(cb), st[i + n + 1]):
                            nxt = cb
                            break
            n += 1
            if nxt is None:
                break
            b = nxt
        gl.append(n)
        i += n
    assert len(gl) == 30 and sum(gl) == len(data)
    _check_replay_consistency(pkg, st, pi, z, np.array(gl), 8, onehot_late=False)


def test_selfplay_full_size_properties(pkg):
    """BASELINE.json configs[1] at full size: 4096 concurrent games, 50 sims/move, 10x128 network, two lanes.
    The oracle cannot replay 12 M network evaluations in a test, so the run is checked through properties that
    do not depend on size: every sample is a visit distribution of exactly 50 simulations over legal moves,
    plane 2 is the oracle's legal mask of planes 0/1 (all samples, vectorised), z follows L16, the counters
    balance, and 96 sampled games are walked move by move with the oracle's rules."""
    torch.manual_seed(42)
    net = pkg.OthelloResNet(10, 128).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=50, temperature_threshold=15,
                                   num_parallel_games=4096, verbose=False)
    np.random.seed(7)
    assert w.lanes == 2     # picked automatically at this size
    st, pi, z = w._run_device(4096, True)
    n = len(z)
    c = {}
    for eng in w._lane_engines:
        for k, v in eng.counters().items():
            c[k] = c.get(k, 0) + v
    assert c["games"] == 4096 and c["plies"] == n and c["simulations"] == 50 * n
    assert c["evals"] == n + c["simulations"] - c["terminal_sims"]
    assert 4096 * 40 < n < 4096 * 75
    # planes are 0/1, disjoint stones, and plane 2 == legal mask computed by the oracle from planes 0/1
    assert bool(((st == 0) | (st == 1)).all())
    assert not np.any((st[:, 0] == 1) & (st[:, 1] == 1))
    weights = (np.uint64(1) << np.arange(64, dtype=U64)).reshape(8, 8)
    sb = (st[:, 0].astype(U64) * weights).sum(axis=(1, 2), dtype=U64)
    ob = (st[:, 1].astype(U64) * weights).sum(axis=(1, 2), dtype=U64)
    lg = (st[:, 2].astype(U64) * weights).sum(axis=(1, 2), dtype=U64)
    assert np.array_equal(ol.legal_batch(sb, ob), lg)
    # pi: exactly 50 visits spread over legal moves (or all 50 on the pass move when there is none)
    v50 = pi.astype(np.float64) * 50.0
    assert np.abs(v50 - np.rint(v50)).max() < 1e-3 and np.abs(v50.sum(axis=1) - 50.0).max() < 1e-3
    legal65 = np.concatenate([st[:, 2].reshape(n, 64) > 0, (lg == 0)[:, None]], axis=1)
    assert not np.any((pi > 0) & ~legal65)
    assert set(np.unique(z).tolist()) <= {-1.0, 0.0, 1.0}
    # game boundaries: a sample equal to the start position with an even-ply successor chain; walk 96 games
    start = ol.tensor(ol.board())
    is_start = np.all(st == start[None], axis=(1, 2, 3))
    firsts = np.flatnonzero(is_start)
    assert len(firsts) == 4096 and firsts[0] == 0
    lens = np.diff(np.append(firsts, n))
    rng = np.random.default_rng(0)
    for g in rng.choice(4096, 96, replace=False):
        a, L = int(firsts[g]), int(lens[g])
        _check_replay_consistency(pkg, st[a:a + L], pi[a:a + L], z[a:a + L], np.array([L]), 15, onehot_late=False)


def test_no_device_memory_growth(pkg, children):
    """Repeated runs of different sizes on one engine, plus engine / evaluator create-destroy cycles, must not
    leak device memory (hipMemGetInfo before and after; the library allocates with hipMalloc, outside torch's
    caching allocator)."""
    children.wait_all()   # the child stages (rehearsal ranks, RCCL checks) share this GPU: device-wide free memory moves with them
    torch.manual_seed(2)
    net = pkg.OthelloResNet(2, 16).eval()
    w = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=5, temperature_threshold=6,
                                   num_parallel_games=64, verbose=False)
    np.random.seed(0)
    w.execute_episodes(200)          # largest run first: output buffers reach their final capacity
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for n in (7, 200, 64, 1, 130, 200):
        assert len(w.execute_episodes(n)) > 9 * n
    for _ in range(5):
        w2 = pkg.ParallelSelfPlayWorker(pkg.OthelloBitboard, net, num_simulations=5, temperature_threshold=6,
                                        num_parallel_games=32, verbose=False, eval_cache_log2=12)
        w2.execute_episodes(40)
        del w2
    import gc
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory shrank by %d bytes" % (free0 - free1)


def test_integration_md_ctypes_stub_runs(pkg):
    """The reference-side ctypes binding printed in INTEGRATION.md (section 2) is executed verbatim against the built
    library: documentation that does not run is wrong documentation."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = [b for b in blocks if "class MI355XSelfPlayWorker" in b]
    assert len(stub) == 1
    lib = os.path.join(root, "othello_reinforcement_learning_test_amd", "libothello_mi355x.so")
    code = stub[0].replace('C.CDLL("libothello_mi355x.so")', "C.CDLL(%r)" % lib)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(4)
    net = pkg.OthelloResNet(2, 128).eval()
    w = ns["MI355XSelfPlayWorker"](net, num_simulations=5, temperature_threshold=6, num_parallel_games=8)
    np.random.seed(1)
    data = w.execute_episodes(6)
    assert len(data) > 6 * 9 and data[0][0].shape == (3, 8, 8) and data[0][1].shape == (65,)
    st = np.stack([d[0] for d in data]); pi = np.stack([d[1] for d in data])
    z = np.array([d[2] for d in data], dtype=np.float32)
    start = ol.tensor(ol.board())
    firsts = [i for i in range(len(st)) if np.array_equal(st[i], start)]
    assert len(firsts) == 6
    _check_replay_consistency(pkg, st, pi, z, np.diff(np.append(firsts, len(st))), 6, onehot_late=False)
