"""Python handles over sections 3 and 4 of the C ABI: the HIP evaluator and the search/self-play
engine.  Used by mcts.py, self_play.py and parallel_self_play.py (the mirrors of the reference
classes); nothing here computes anything on the host.
"""
import ctypes as C

import numpy as np

from . import _lib


def default_precision(num_filters, board_size=8):
    """fp16-split MFMA trunks (fp32-equivalent: three f16 products per operand pair) where one exists -- 128 filters on
    8x8 (k_trunk16, the benchmarked kernel), 32 / 64 filters on 8x8 and 6x6 and 128 filters on 6x6 (k_trunk_h3) -- else
    the exact-fp32 MFMA trunk (16 filters)."""
    if num_filters == 128:
        return "f16x3"     # 8x8: k_trunk16; 6x6: k_trunk_h3
    return "f16x3" if num_filters in (32, 64) else "f32"


_PROBE = {}   # board size -> (self, opp, legal) uint64 arrays


def probe_positions(board_size=8, n=256):
    """A fixed set of n positions of seeded random games (its own numpy Generator: the global RNG the workers draw their seeds
    from is not touched), played with the library's host rules: what HipResNetEvaluator.refresh() evaluates once after a weight
    update to choose the activation scale as a function of the WEIGHTS (see refresh)."""
    got = _PROBE.get((board_size, n))
    if got is None:
        lib = _lib.load()
        rng = np.random.Generator(np.random.PCG64(20240607 + board_size))
        pos = []
        while len(pos) < n:
            b = _lib.Board()
            lib.oth_board_reset_n(board_size, C.byref(b))
            while not lib.oth_board_is_terminal_n(board_size, C.byref(b)) and len(pos) < n:
                legal = int(lib.oth_legal_moves_n(board_size, b.self_board, b.opp_board))
                pos.append((b.self_board, b.opp_board, legal))
                moves = [a for a in range(board_size * board_size) if (legal >> a) & 1] or [board_size * board_size]
                lib.oth_board_make_move_n(board_size, C.byref(b), int(moves[int(rng.integers(len(moves)))]))
        got = tuple(np.array([p[j] for p in pos], dtype=np.uint64) for j in range(3))
        _PROBE[(board_size, n)] = got
    return got


class HipResNetEvaluator:
    """OthelloResNet forward (reference net.py:182-205, eval mode) on the GPU.

    ``model`` is any module with the reference's state_dict layout (the reference's own
    OthelloResNet or net.OthelloResNet).  The trainer keeps updating that module between
    ``execute_episodes`` calls (SURVEY 8(b) threading note), so ``refresh()`` re-reads the weights
    whenever a parameter/buffer version counter has moved.
    """

    def __init__(self, model, precision=None):
        _lib.require_device()
        self.model = model
        self.num_blocks = int(getattr(model, "num_blocks", len(model.res_blocks)))
        self.num_filters = int(getattr(model, "num_filters", model.conv_block.conv.out_channels))
        board_size = int(getattr(model, "board_size", 8))
        self.board_size = board_size
        self.policy_size = board_size * board_size + 1
        self.precision = precision or default_precision(self.num_filters, board_size)
        self._precision0 = self.precision   # what a weight update goes back to (a rescue may have switched to "f32")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self._h = _lib.load().oth_net_create(self.num_blocks, self.num_filters, board_size)
        if not self._h:
            raise _lib.OthelloHipError("oth_net_create: " + _lib.last_error())
        self._version = None
        # rescue of a saturated launch (see needs_rescue): how often it happened, and what was done about it
        self.rescues = []
        self.refresh()

    def _model_version(self):
        v = 0
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            v += int(t._version) + 1
        return v

    def refresh(self, force=False):
        """Re-upload (fold + repack) the weights if the model changed since the last call.

        A weight UPDATE (the version counters moved since the last upload: the trainer stepped, trainer.py:283-328) also
        forgets what earlier weights did to the activation scale and the precision (the scale used to be sticky,
        so results depended on the process's history): both go back to their starting values (16, the constructor's
        precision), the new weights are uploaded, and ONE launch over a fixed set of 256 probe positions (probe_positions)
        lets the rescue lower the scale again if these weights need it -- so the scale a call runs at is a function of the
        weights (and, only if a position outside the probe set saturates later, of that call).  No launch of this network
        may be in flight (the workers call refresh() at the start of a call)."""
        ver = self._model_version()
        moved = self._version is not None and ver != self._version
        if force or ver != self._version:
            if moved and not force and self._precision0 != "f32":
                self.precision = self._precision0
                _lib.call("oth_net_set_act_scale", self._h, C.c_float(16.0))
            blob = _lib.state_dict_blob(self.model.state_dict())
            _lib.call("oth_net_load_state", self._h, _lib.np_ptr(blob, C.c_float), blob.size,
                      _lib.PRECISIONS[self.precision])
            self._version = ver
            if moved and not force and self.precision != "f32":
                self.calibrate()

    def calibrate(self):
        """One rescued launch over the probe positions: -> the activation scale these weights run at."""
        import torch
        sb, ob, lg = (torch.from_numpy(a.view(np.int64)).cuda() for a in probe_positions(self.board_size))
        self.forward_bits(sb, ob, lg, rescue=True)
        return self.act_scale

    @property
    def handle(self):
        return self._h

    def forward_planes(self, x, rescue=True):
        """x: CUDA float32 (N,3,S,S) of 0/1 planes -> (log-probs (N,S*S+1), value (N,1)) CUDA tensors.

        ``rescue=True`` (default): an fp16-split launch that clamped an activation is run again at a lower activation
        scale (``needs_rescue``).  THE CHECK SYNCHRONISES: every call then ends with a 4-byte device-to-host copy and a
        hipStreamSynchronize, i.e. host and device meet once per call -- right for a caller that reads the outputs next
        (``.cpu()`` synchronises anyway), wrong inside a pipeline of launches.  ``rescue=False`` launches once,
        ASYNCHRONOUSLY, and leaves the device flag for the caller to collect at its own join point (``needs_rescue`` and
        run the work again, or ``check_saturation`` to fail loudly) -- what the workers and the engine do."""
        import torch
        x = x.contiguous()
        n = x.shape[0]
        if tuple(x.shape[1:]) != (3, self.board_size, self.board_size):
            raise ValueError("expected input of shape (N,3,%d,%d)" % (self.board_size, self.board_size))
        logp = torch.empty((n, self.policy_size), dtype=torch.float32, device=x.device)
        v = torch.empty((n,), dtype=torch.float32, device=x.device)
        while True:
            _lib.call("oth_net_forward_planes", self._h, x.data_ptr(), n, logp.data_ptr(), v.data_ptr(),
                      _lib.current_stream())
            if not (rescue and self.needs_rescue()):
                return logp, v.view(n, 1)

    def forward_bits(self, self_b, opp_b, legal, rescue=True):
        """Packed-bitboard input (CUDA int64 tensors) -> (log-probs, value); ``rescue`` as in forward_planes."""
        import torch
        n = self_b.numel()
        logp = torch.empty((n, self.policy_size), dtype=torch.float32, device=self_b.device)
        v = torch.empty((n,), dtype=torch.float32, device=self_b.device)
        while True:
            _lib.call("oth_net_forward_bits", self._h, self_b.data_ptr(), opp_b.data_ptr(), legal.data_ptr(),
                      n, None, logp.data_ptr(), v.data_ptr(), _lib.current_stream())
            if not (rescue and self.needs_rescue()):
                return logp, v.view(n, 1)

    def kernel_info(self, n_positions=4096):
        """Which trunk kernel a launch of ``n_positions`` runs, as the library itself dispatches it:
        {"kernel": name, "issued_per_flop": MFMA FLOPs issued per algorithmic FLOP, "clamp": activation clamp or 0}."""
        name = C.create_string_buffer(256)
        issued, clamp = C.c_double(0), C.c_double(0)
        _lib.call("oth_net_kernel_info", self._h, int(n_positions), name, 256, C.byref(issued), C.byref(clamp))
        return {"kernel": name.value.decode(), "issued_per_flop": issued.value, "clamp": clamp.value,
                "act_scale": self.act_scale}

    def saturated(self):
        """True when an fp16-split trunk launch since the last call clamped an activation (at 60000 / act_scale in the
        direct kernels, 30000 / act_scale in the Winograd trunks that are the default for 10x128 on 8x8 and 5x64 on 6x6:
        3750 / 1875 at the default scale 16; the reference's fp32 forward has no clamp): reads and clears the device flag
        (one 4-byte copy, synchronises the stream)."""
        flag = C.c_int32(0)
        _lib.call("oth_net_saturated", self._h, C.byref(flag), _lib.current_stream())
        return bool(flag.value)

    @property
    def act_scale(self):
        """Power-of-two pre-scale of the activations in the fp16-split trunks (16 by default, lowered by needs_rescue; the
        exact-fp32 trunk of precision 'f32' does not use it)."""
        s = C.c_float(0)
        _lib.call("oth_net_get_act_scale", self._h, C.byref(s))
        return float(s.value)

    def needs_rescue(self):
        """The reference's fp32 forward has no clamp (net.py:182-205), so a launch that clamped must not stand.  Called
        by every worker / search mirror when a call's launches have ended (no launch of this network may be in flight):
        False = nothing clamped, the results stand.  True = something did, and the evaluator has widened its range --
        the activation scale halved (16 -> 8 -> 4 -> 2 -> 1: 1875 -> 30 000 in the Winograd trunks), or, if scale 1 still
        clamped, the weights repacked for the exact-fp32 MFMA trunk -- so the caller must run the call again from the
        state it started in (SearchEngine.snapshot / restore for a stream step or a lock-step search; a batch run
        restarts from its seed; a search from its roots).  Seeded-random networks never get here; every rescue is
        recorded in ``self.rescues`` and announced once per step with a warning."""
        if self.precision == "f32" or not self.saturated():
            return False
        import warnings
        info = self.kernel_info()
        scale = self.act_scale
        if scale > 1.0:
            _lib.call("oth_net_set_act_scale", self._h, C.c_float(scale / 2))
            what = "activation scale %g -> %g (clamp %g -> %g)" % (scale, scale / 2, info["clamp"], 2 * info["clamp"])
        else:
            self.precision = "f32"
            self.refresh(force=True)
            what = "still clamped at activation scale 1 (clamp %g): weights repacked for the exact-fp32 MFMA trunk" % info["clamp"]
        self.saturated()   # (clear a flag raised by launches that were still in the call's tail)
        self.rescues.append(what)
        warnings.warn("an activation of the %dx%d network reached the range of the fp16-split trunk kernel %s; %s; the "
                      "affected call is run again" % (self.num_blocks, self.num_filters, info["kernel"].split(" ")[0], what),
                      RuntimeWarning, stacklevel=3)
        return True

    def check_saturation(self):
        """Last line of defence (loud failure instead of a silent deviation from the reference): raises if a launch
        clamped an activation and nobody rescued the call.  The workers and search mirrors use needs_rescue() and run the
        call again; this is for callers that launch with rescue=False or drive the C ABI themselves."""
        if self.precision != "f32" and self.saturated():
            info = self.kernel_info()
            raise _lib.OthelloHipError(
                "an activation of the %dx%d network exceeded %g, the range of the fp16-split trunk kernel %s at activation "
                "scale %g: results would differ from the reference's fp32 forward -- run the call again after "
                "needs_rescue(), or build the evaluator with precision='f32'"
                % (self.num_blocks, self.num_filters, info["clamp"], info["kernel"].split(" ")[0], self.act_scale))

    def policy_probs(self, logp):
        """exp(log-probs) with the engine's own expf (what the expansion feeds node.py:71-80): CUDA tensor in/out."""
        import torch
        logp = logp.contiguous()
        out = torch.empty_like(logp)
        _lib.call("oth_policy_exp", logp.data_ptr(), out.data_ptr(), logp.numel(), _lib.current_stream())
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.load().oth_net_destroy(self._h)
                self._h = None
        except Exception:
            pass


_LANE_STREAMS = {}   # device index -> {"pool": [torch.cuda.Stream, ...], "start": index of the first stream in use}


def lane_streams(n, device=None, redraw=False):
    """The streams of the n lanes of a multi-lane run on `device`: made ONCE per process and device and reused by every run.

    Not a micro-optimisation.  HIP maps streams onto a few hardware queues; with new streams per run the two streams of every
    SECOND two-lane run of a process landed on ONE queue, their trunk launches alternated instead of overlapping and the run
    lost 8-10 % (round 5: three identical two-lane legs in one process measured 1 682 / 1 518 / 1 675 games/s with new streams
    per leg, 1 681 / 1 681 / 1 680 with these; profiles/r05_lane_modes.log).  The first streams of a process get distinct
    queues, so the streams are created once and kept.

    ``redraw=True`` (what lane_overlap_check does when it finds the lanes serialised): n NEW streams are appended to the pool
    and become the arrangement every later call returns (``lane_streams_select`` goes back to an earlier one).

    One multi-lane worker at a time per device: every worker of the process is handed these same streams, so two multi-lane
    workers driven concurrently from two host threads would share them and serialise."""
    import torch
    dev = torch.cuda.current_device() if device is None else int(device)
    ent = _LANE_STREAMS.setdefault(dev, {"pool": [], "start": 0})
    if redraw:
        ent["start"] = len(ent["pool"])
    pool, start = ent["pool"], ent["start"]
    while len(pool) < start + n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[start:start + n]


def lane_streams_select(start, device=None):
    """Make the arrangement that begins at pool index `start` the current one (after a redraw that did not help);
    -> the previous start."""
    import torch
    dev = torch.cuda.current_device() if device is None else int(device)
    ent = _LANE_STREAMS.setdefault(dev, {"pool": [], "start": 0})
    old, ent["start"] = ent["start"], int(start)
    return old


def lane_streams_start(device=None):
    import torch
    dev = torch.cuda.current_device() if device is None else int(device)
    return _LANE_STREAMS.setdefault(dev, {"pool": [], "start": 0})["start"]


def union_ms(spans):
    """Total length of the union of (start, end) intervals (ms); `spans` = list of (n, 2) arrays."""
    spans = [s for s in spans if len(s)]
    if not spans:
        return 0.0
    sp = np.concatenate(spans)
    sp = sp[np.argsort(sp[:, 0])]
    total, (cur_s, cur_e) = 0.0, sp[0]
    for s_, e2 in sp[1:]:
        if s_ > cur_e:
            total += cur_e - cur_s
            cur_s, cur_e = s_, e2
        else:
            cur_e = max(cur_e, e2)
    return float(total + (cur_e - cur_s))


# Below these a multi-lane run counts as SERIALISED (its lanes' network launches alternate instead of overlapping -- two
# streams on one hardware queue): sum of launch durations / union of their intervals.  Measured on MI355X
# (profiles/r06_lane_overlap.log): two lanes 1.8 on separate queues and 1.0 on one stream.  The floor grows with the lane
# count (1 + 0.3 per extra lane: 1.3 / 1.6 / 1.9) because with four lanes on four of which two share a queue the figure only
# drops to ~2.  Launches shorter than MIN_LAUNCH_MS are launch-bound (toy networks): the host cannot keep two queues fed,
# the ratio says nothing about the queues and nothing is flagged.
OVERLAP_MIN_LAUNCH_MS = 0.25
OVERLAP_MIN_LAUNCHES = 100      # per lane, before LaneOverlapCheck decides: a 2-ply-round step (22 launches per lane) is too noisy


def overlap_floor(lanes):
    return 1.0 + 0.3 * (max(1, int(lanes)) - 1)


def lanes_overlap(engines, carry=None):
    """-> dict(overlap, sum_ms, union_ms, launches, mean_launch_ms, lanes, serialised) from the HIP-event spans the engines
    recorded during their last run / step (SearchEngine.set_timing(True) before it).  overlap = sum of the network launch
    durations / union of their intervals on the process-wide time axis: ~1 when the lanes' launches alternate, -> the lane
    count when every lane always has a launch running.  `carry`: an earlier result of the same arrangement to add this step to."""
    spans = [e.net_spans() for e in engines]
    tot = float(sum(float((s[:, 1] - s[:, 0]).sum()) for s in spans if len(s)))
    n = int(sum(len(s) for s in spans))
    uni = union_ms(spans)
    if carry:
        tot, uni, n = tot + carry["sum_ms"], uni + carry["union_ms"], n + carry["launches"]
    ov = tot / uni if uni > 0 else 0.0
    mean = tot / n if n else 0.0
    lanes = len(engines)
    return {"overlap": round(ov, 3), "sum_ms": round(tot, 2), "union_ms": round(uni, 2), "launches": n,
            "mean_launch_ms": round(mean, 4), "lanes": lanes, "floor": overlap_floor(lanes),
            "serialised": bool(lanes > 1 and n >= 8 * lanes and mean >= OVERLAP_MIN_LAUNCH_MS and ov < overlap_floor(lanes))}


_WARNED = set()


def warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class LaneOverlapCheck:
    """Lane overlap as a CHECKED property of a multi-lane run (bench.py's Workload, ParallelSelfPlayWorker with lanes > 1).

    Protocol: ``begin(engines)`` before a run / step turns the HIP-event hooks of the engines on; ``end(engines)`` after it
    turns them off, computes ``lanes_overlap`` and decides: a serialised run is reported once on stderr (RuntimeWarning),
    and -- while redraws are left -- the lanes' streams are drawn once more from a widened pool (``lane_streams(redraw=
    True)``) so that the caller's NEXT run is measured on the new arrangement; the better of the arrangements seen is kept.
    ``pending`` says whether the next run should be measured too (also while fewer than ``min_launches`` launches per lane have
    been seen on the current arrangement: the decision is taken on the accumulated sample).  ``report()`` -> what bench.py /
    last_stats print."""

    def __init__(self, lanes, device=None, max_redraws=1, min_launches=OVERLAP_MIN_LAUNCHES):
        self.lanes, self.device = int(lanes), device
        self.max_redraws = int(max_redraws)
        self.min_launches = int(min_launches)
        self._carry = None
        self.history = []            # [(pool start, lanes_overlap dict)]
        self.pending = self.lanes > 1
        self.redraws = 0
        self.kept = None
        self._was = []
        at = _lib.RUNTIME_AT_IMPORT
        if self.lanes > 3 and at["hip_runtime_up_before_import"] and at["GPU_MAX_HW_QUEUES_from_caller"] is None:
            warn_once("late-queues", "%d lanes, but the HIP runtime was started before this package was imported, so its "
                      "GPU_MAX_HW_QUEUES=8 default has no effect (HIP's own default is 4 hardware queues, one of them the "
                      "default stream's): two lanes will share a queue and serialise (-8 %% measured on BASELINE configs[4]).  "
                      "Import the package, or set GPU_MAX_HW_QUEUES=8, before the first GPU call" % self.lanes)

    def begin(self, engines):
        self._was = [e.timing for e in engines]
        for e in engines:
            e.set_timing(True)

    def end(self, engines):
        """-> True when the caller must fetch its streams again (lane_streams) before the next run."""
        m = lanes_overlap(engines, self._carry)
        for e, was in zip(engines, self._was):
            e.set_timing(was)
        if m["launches"] < self.min_launches * self.lanes and m["mean_launch_ms"] >= OVERLAP_MIN_LAUNCH_MS:
            self._carry = m          # too small a sample to decide on: measure the next run as well
            return False
        self._carry = None
        start = lane_streams_start(self.device)
        self.history.append((start, m))
        self.pending = False
        if not m["serialised"]:
            if self.kept is None:
                self.kept = len(self.history) - 1
            elif m["overlap"] > self.history[self.kept][1]["overlap"]:
                self.kept = len(self.history) - 1
            if self.kept != len(self.history) - 1:
                lane_streams_select(self.history[self.kept][0], self.device)
                return True
            return False
        late = _lib.RUNTIME_AT_IMPORT["hip_runtime_up_before_import"] and \
            _lib.RUNTIME_AT_IMPORT["GPU_MAX_HW_QUEUES_from_caller"] is None
        warn_once(("serialised", self.lanes),
                  "the %d lanes' network launches do not overlap (sum / union of the launch intervals %.2f, expected >= %.1f; "
                  "mean launch %.2f ms): their streams share a hardware queue and the run loses 8-10 %%.  GPU_MAX_HW_QUEUES=%s%s"
                  % (self.lanes, m["overlap"], m["floor"], m["mean_launch_ms"], _lib.runtime_env()["GPU_MAX_HW_QUEUES"],
                     " was set by this package AFTER the HIP runtime had started, so it has no effect: import the package (or "
                     "set GPU_MAX_HW_QUEUES=8) before the first GPU call" if late else ""))
        if self.redraws < self.max_redraws:
            self.redraws += 1
            lane_streams(self.lanes, self.device, redraw=True)
            self.pending = True
            return True
        # out of redraws: keep the best arrangement seen
        best = max(range(len(self.history)), key=lambda i: self.history[i][1]["overlap"])
        self.kept = best
        if self.history[best][0] != start:
            lane_streams_select(self.history[best][0], self.device)
            return True
        return False

    def report(self):
        if not self.history:
            if self._carry:
                return {"lanes_overlap": self._carry["overlap"], "lanes_serialised": False, "undecided_after_launches": self._carry["launches"]}
            return {"lanes_overlap": None, "lanes_serialised": None}
        kept = self.history[self.kept if self.kept is not None else -1][1]
        return {"lanes_overlap": kept["overlap"], "lanes_serialised": kept["serialised"], "floor": kept["floor"],
                "mean_launch_ms": kept["mean_launch_ms"], "stream_redraws": self.redraws,
                "arrangements_tried": [h[1]["overlap"] for h in self.history]}


def policy_from_visits(visits, self_b, opp_b, temperature, board_size=8):
    """MCTSNode.get_policy_distribution for a general temperature (/root/reference/src/mcts/node.py:162-182), the same
    numpy expressions in the same order on the root's children (= the legal moves in ascending order, or the pass):
    float32 counts ** (1.0 / T), /= counts.sum(), scatter.  Bit-identical to the reference by construction.

    A TERMINAL root is not node.py:164's childless case: get_legal_moves() returns [64] there (bitboard.pyx:177-185), the
    root is expanded with the pass child and every simulation is backed up through it, so the policy is one-hot on the
    pass at every temperature (fixture g9, generated by the reference).  Only a search of zero simulations has all-zero
    counts: at T = 0 the reference then returns the one-hot on actions[argmax(zeros)] = the FIRST child (node.py:170-173)
    and so does this mirror; at T != 0 the reference evaluates 0/0 and this mirror returns the zero vector instead of NaN
    (a deliberate deviation; parity unpinned: no fixture has zero simulations)."""
    npol = board_size * board_size + 1
    policy = np.zeros(npol, dtype=np.float32)
    legal = int(_lib.load().oth_legal_moves_n(board_size, self_b, opp_b))
    actions = [a for a in range(npol - 1) if (legal >> a) & 1] or [npol - 1]
    counts = np.array([visits[a] for a in actions], dtype=np.float32)
    if temperature != 0 and not counts.any():
        return policy
    if temperature == 0:
        policy[actions[int(np.argmax(counts))]] = 1.0
    else:
        counts = counts ** (1.0 / temperature)
        counts /= counts.sum()
        for a, p in zip(actions, counts):
            policy[a] = p
    return policy


class SearchEngine:
    """Handle over oth_engine: G concurrent game slots, one wavefront per game on the device."""

    def __init__(self, max_games, num_simulations, temperature_threshold=15, c_puct=1.0,
                 dirichlet_alpha=0.3, dirichlet_epsilon=0.25, store_late_onehot=False, evaluator=None,
                 eval_cache_log2=0, board_size=None):
        _lib.require_device()
        self.max_games = int(max_games)
        self.num_simulations = int(num_simulations)
        if board_size is None:   # follow the evaluator's network; 8 (the reference's game) otherwise
            board_size = getattr(evaluator, "board_size", 8) if evaluator is not None else 8
        self.board_size = int(board_size)
        self.cells = self.board_size * self.board_size
        self.npol = self.cells + 1
        cfg = _lib.EngineCfg(self.max_games, self.num_simulations, int(temperature_threshold),
                             float(c_puct), float(dirichlet_alpha), float(dirichlet_epsilon),
                             1 if store_late_onehot else 0, int(eval_cache_log2), self.board_size)
        self._h = _lib.load().oth_engine_create(C.byref(cfg))
        if not self._h:
            raise _lib.OthelloHipError("oth_engine_create: " + _lib.last_error())
        self.evaluator = None
        if evaluator is not None:
            self.set_evaluator(evaluator)

    def set_evaluator(self, evaluator):
        self.evaluator = evaluator
        _lib.call("oth_engine_set_net", self._h, evaluator.handle)

    # ---- step-wise search ------------------------------------------------------------------
    def search_begin(self, self_b, opp_b):
        s = np.ascontiguousarray(self_b, dtype=np.uint64)
        o = np.ascontiguousarray(opp_b, dtype=np.uint64)
        self._n = len(s)
        self._roots = (s.copy(), o.copy())
        _lib.call("oth_search_begin", self._h, _lib.np_ptr(s, C.c_uint64), _lib.np_ptr(o, C.c_uint64),
                  self._n, _lib.current_stream())

    def search_select(self):
        _lib.call("oth_search_select", self._h, _lib.current_stream())

    def search_leaves(self):
        cnt = C.c_int32(0)
        s = np.zeros(self.max_games, dtype=np.uint64)
        o = np.zeros(self.max_games, dtype=np.uint64)
        lg = np.zeros(self.max_games, dtype=np.uint64)
        _lib.call("oth_search_leaves", self._h, C.byref(cnt), _lib.np_ptr(s, C.c_uint64),
                  _lib.np_ptr(o, C.c_uint64), _lib.np_ptr(lg, C.c_uint64), _lib.current_stream())
        n = cnt.value
        return s[:n], o[:n], lg[:n]

    def search_expand(self, policy, value, is_log=False):
        p = np.ascontiguousarray(policy, dtype=np.float32)
        v = np.ascontiguousarray(value, dtype=np.float32)
        _lib.call("oth_search_expand", self._h, p.ctypes.data, v.ctypes.data, 1 if is_log else 0,
                  _lib.current_stream())

    def search_run(self):
        _lib.call("oth_search_run", self._h, _lib.current_stream())

    def search_results(self, temperature=1.0):
        """-> (pi, visits, value_sum, prior) of the roots.  temperature 0 / 1 come from the device (k_results); any
        other value is node.py:175-177 evaluated literally on the host from the device's visit counts."""
        if temperature not in (0, 0.0, 1, 1.0):
            _, visits, wsum, prior = self.search_results(1.0)
            pi = np.stack([policy_from_visits(visits[i], int(self._roots[0][i]), int(self._roots[1][i]), temperature,
                                              self.board_size) for i in range(self._n)])
            return pi, visits, wsum, prior
        n = self._n
        pi = np.zeros((n, self.npol), dtype=np.float32)
        visits = np.zeros((n, self.npol), dtype=np.int32)
        wsum = np.zeros((n, self.npol), dtype=np.float64)
        prior = np.zeros((n, self.npol), dtype=np.float32)
        _lib.call("oth_search_results", self._h, float(temperature), _lib.np_ptr(pi, C.c_float),
                  _lib.np_ptr(visits, C.c_int32), _lib.np_ptr(wsum, C.c_double),
                  _lib.np_ptr(prior, C.c_float), _lib.current_stream())
        return pi, visits, wsum, prior

    def search_with(self, self_b, opp_b, eval_fn, temperature=1.0):
        """Whole search with an external evaluator: eval_fn(self u64[m], opp u64[m], legal u64[m])
        -> (probs f32[m,65], values f32[m]).  Mirrors BatchMCTS.search_batch."""
        self.search_begin(self_b, opp_b)
        s, o, lg = self.search_leaves()
        p, v = eval_fn(s, o, lg)
        self.search_expand(p, v, is_log=False)
        for _ in range(self.num_simulations):
            self.search_select()
            s, o, lg = self.search_leaves()
            if len(s):
                p, v = eval_fn(s, o, lg)
            else:
                p, v = np.zeros((1, self.npol), np.float32), np.zeros(1, np.float32)
            self.search_expand(p, v, is_log=False)
        return self.search_results(temperature)

    # ---- self-play -------------------------------------------------------------------------
    def selfplay_run(self, num_games, seed, add_noise=True):
        n = C.c_int64(0)
        _lib.call("oth_selfplay_run", self._h, int(num_games), C.c_uint64(int(seed) & (2**64 - 1)),
                  1 if add_noise else 0, C.byref(n), _lib.current_stream())
        self._run_games = int(num_games)
        return n.value

    # ---- streaming self-play: the slots stay full across steps ------------------------------
    def stream_begin(self, seed, stagger_rounds=0, hist_games=0):
        _lib.call("oth_stream_begin", self._h, C.c_uint64(int(seed) & (2**64 - 1)), int(stagger_rounds),
                  int(hist_games), _lib.current_stream())

    def stream_step(self, min_games):
        """Play until >= min_games more games have finished; -> (games, samples) of this step's harvest."""
        g, n = C.c_int32(0), C.c_int64(0)
        _lib.call("oth_stream_step", self._h, int(min_games), C.byref(g), C.byref(n), _lib.current_stream())
        self._run_games = g.value
        return g.value, n.value

    def snapshot(self):
        """Keep the state of the stream / lock-step run as it is NOW (between two calls) ..."""
        _lib.call("oth_engine_snapshot", self._h, _lib.current_stream())

    def restore(self):
        """... and put it back: the step / search since the snapshot is forgotten and can be played again (the rescue of
        a saturated fp16-split launch, HipResNetEvaluator.needs_rescue)."""
        _lib.call("oth_engine_restore", self._h, _lib.current_stream())

    def _rescuable(self):
        ev = self.evaluator
        return ev is not None and getattr(ev, "precision", "f32") != "f32" and hasattr(ev, "needs_rescue")

    def search_run_rescued(self):
        """search_run from the roots of the last search_begin, repeated while a launch saturated."""
        while True:
            self.search_run()
            if not (self._rescuable() and self.evaluator.needs_rescue()):
                return
            self.search_begin(*self._roots)

    def selfplay_run_rescued(self, num_games, seed, add_noise=True):
        """selfplay_run, restarted from its seed while a launch saturated (a run is a pure function of the seed)."""
        while True:
            n = self.selfplay_run(num_games, seed, add_noise)
            if not (self._rescuable() and self.evaluator.needs_rescue()):
                return n

    def stream_step_rescued(self, min_games):
        """stream_step from a snapshot, replayed while a launch saturated.  One engine per evaluator only: with several
        lanes sharing an evaluator the snapshot / check / restore belongs at the point where all lanes have joined."""
        while True:
            if self._rescuable():
                self.snapshot()
            out = self.stream_step(min_games)
            if not (self._rescuable() and self.evaluator.needs_rescue()):
                return out
            self.restore()

    def selfplay_search_rescued(self):
        while True:
            if self._rescuable():
                self.snapshot()
            out = self.selfplay_search()
            if not (self._rescuable() and self.evaluator.needs_rescue()):
                return out
            self.restore()

    def game_ids(self):
        """ids of the games of the last run / step, in output order."""
        cnt = C.c_int32(0)
        _lib.call("oth_selfplay_game_ids", self._h, None, 0, C.byref(cnt))
        ids = np.zeros(cnt.value, dtype=np.int32)
        if cnt.value:
            _lib.call("oth_selfplay_game_ids", self._h, _lib.np_ptr(ids, C.c_int32), cnt.value, C.byref(cnt))
        return ids

    def selfplay_begin(self, n):
        self._n = int(n)
        self._run_games = int(n)
        _lib.call("oth_selfplay_begin", self._h, self._n, _lib.current_stream())

    def selfplay_search(self):
        pi = np.zeros((self._n, self.npol), dtype=np.float32)
        active = np.zeros(self._n, dtype=np.int32)
        _lib.call("oth_selfplay_search", self._h, _lib.np_ptr(pi, C.c_float),
                  _lib.np_ptr(active, C.c_int32), _lib.current_stream())
        return pi, active

    def selfplay_apply(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        left = C.c_int32(0)
        _lib.call("oth_selfplay_apply", self._h, _lib.np_ptr(a, C.c_int32), C.byref(left),
                  _lib.current_stream())
        return left.value

    def selfplay_end(self):
        n = C.c_int64(0)
        _lib.call("oth_selfplay_end", self._h, C.byref(n), _lib.current_stream())
        return n.value

    def selfplay_fetch(self, n_samples):
        """-> host numpy arrays (states (n,3,S,S), pis (n,S*S+1), zs (n,), game_len (games,))"""
        st = np.empty((n_samples, 3, self.board_size, self.board_size), dtype=np.float32)
        pi = np.empty((n_samples, self.npol), dtype=np.float32)
        z = np.empty((n_samples,), dtype=np.float32)
        gl = np.zeros((self._run_games,), dtype=np.int32)
        _lib.call("oth_selfplay_fetch", self._h, st.ctypes.data, pi.ctypes.data, z.ctypes.data,
                  gl.ctypes.data, _lib.current_stream())
        return st, pi, z, gl

    def selfplay_device_tensors(self):
        """Zero-copy CUDA views of the compacted replay tuples of the last run (valid until the
        next run): for on-device consumers such as the RCCL all-gather."""
        import torch
        ps, pp, pz, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(0)
        _lib.call("oth_selfplay_device_ptrs", self._h, C.byref(ps), C.byref(pp), C.byref(pz), C.byref(n))
        n = n.value

        def view(ptr, shape):
            if n == 0:
                return torch.empty(shape, dtype=torch.float32, device="cuda")
            numel = int(np.prod(shape))
            iface = {"shape": (numel,), "typestr": "<f4", "data": (ptr.value, False), "version": 2}
            holder = type("_DevMem", (), {"__cuda_array_interface__": iface})()
            return torch.as_tensor(holder, device="cuda").view(*shape)
        return view(ps, (n, 3, self.board_size, self.board_size)), view(pp, (n, self.npol)), view(pz, (n,))

    def counters(self):
        out = (C.c_int64 * 8)()
        _lib.call("oth_engine_counters", self._h, out)
        keys = ("evals", "simulations", "plies", "games", "net_batches", "terminal_sims", "cache_hits")
        return dict(zip(keys, list(out)[:7]))

    def cache_stats(self):
        """Evaluation-cache statistics since the run / stream began: distinct positions evaluated (compulsory misses),
        repeated evaluations (the entry was replaced in between, or a double miss in one launch), conflict evictions,
        entries of the table."""
        out = (C.c_int64 * 4)()
        _lib.call("oth_engine_cache_stats", self._h, out, _lib.current_stream())
        return dict(zip(("distinct_positions", "repeated_evals", "conflict_evictions", "entries"), list(out)))

    timing = False   # HIP-event hooks on every launch (set_timing)

    def set_timing(self, enable=True):
        _lib.call("oth_engine_set_timing", self._h, 1 if enable else 0)
        self.timing = bool(enable)

    def net_spans(self):
        """(n, 2) array of (start, end) ms of every network launch of the last run on a process-wide time axis
        (needs set_timing(True))."""
        n = C.c_int64(0)
        _lib.call("oth_engine_net_spans", self._h, None, 0, C.byref(n))
        out = np.zeros((n.value, 2), dtype=np.float64)
        if n.value:
            _lib.call("oth_engine_net_spans", self._h, _lib.np_ptr(out, C.c_double), n.value, C.byref(n))
        return out

    def kernel_time(self):
        nm, tm = C.c_double(0), C.c_double(0)
        nl, tl = C.c_int64(0), C.c_int64(0)
        _lib.call("oth_engine_kernel_time", self._h, C.byref(nm), C.byref(nl), C.byref(tm), C.byref(tl))
        return {"net_ms": nm.value, "net_launches": nl.value, "tree_ms": tm.value, "tree_launches": tl.value}

    def __del__(self):
        try:
            if self._h:
                _lib.load().oth_engine_destroy(self._h)
                self._h = None
        except Exception:
            pass
