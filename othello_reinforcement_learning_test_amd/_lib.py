"""ctypes binding of libothello_mi355x.so (include/othello_mi355x.h).

The library is built in-tree (othello_reinforcement_learning_test_amd/csrc/Makefile ->
othello_reinforcement_learning_test_amd/libothello_mi355x.so).  There is no fallback of any kind:
a missing library raises ImportError here, and every device entry point raises ``OthelloHipError``
when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one taken by the default stream); multi-lane runs put one
# stream per lane to work, and two lanes on one queue serialise their launches (four lanes: 25.7 k instead of 28 k games/s on
# BASELINE configs[4], profiles/r05_hw_queues.log).  Read when the HIP runtime starts, so this only helps when the package is
# imported before the first GPU call; a caller's own setting wins.
def _hip_runtime_up():
    """True when this process had already started the HIP runtime through torch when the package was imported (the one case
    this module can see; a HIP call made by other native code before the import is invisible to it)."""
    import sys
    torch = sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


# What the two runtime knobs were when the package was imported, for the operator and for bench.py's JSON line
# (`runtime_env`): GPU_MAX_HW_QUEUES as the caller had it (None = unset, this module then sets 8) and whether the HIP runtime was
# already up -- in which case the setting below comes too late and lanes > 3 share hardware queues (lane_overlap_check warns).
RUNTIME_AT_IMPORT = {"GPU_MAX_HW_QUEUES_from_caller": os.environ.get("GPU_MAX_HW_QUEUES"),
                     "hip_runtime_up_before_import": _hip_runtime_up()}
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def runtime_env():
    """The effective values of the environment knobs the multi-lane / multi-rank paths depend on."""
    return {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
            "GPU_MAX_HW_QUEUES_set_by": ("caller" if RUNTIME_AT_IMPORT["GPU_MAX_HW_QUEUES_from_caller"] is not None
                                         else "package default (8)"),
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "package_imported_before_hip_runtime": not RUNTIME_AT_IMPORT["hip_runtime_up_before_import"]}

_HERE = os.path.dirname(os.path.abspath(__file__))
# OTHELLO_MI355X_LIB: A/B-testing hook for kernel work (another build of the same library)
LIB_PATH = os.environ.get("OTHELLO_MI355X_LIB") or os.path.join(_HERE, "libothello_mi355x.so")

OTH_PREC_F32, OTH_PREC_F16X3, OTH_PREC_F16, OTH_PREC_F16X3_DIRECT = 0, 1, 2, 3
# "f16x3": the default for 32 / 64 / 128 filters (128 on 8x8: the 1-D Winograd trunk); "f16x3_direct": the same arithmetic
# on the direct-convolution kernel (128 filters on 8x8 only)
PRECISIONS = {"f32": OTH_PREC_F32, "f16x3": OTH_PREC_F16X3, "f16": OTH_PREC_F16, "f16x3_direct": OTH_PREC_F16X3_DIRECT}


class OthelloHipError(RuntimeError):
    """An entry point of libothello_mi355x.so reported a failure."""


class Board(C.Structure):  # oth_board
    _fields_ = [("self_board", C.c_uint64), ("opp_board", C.c_uint64),
                ("move_count", C.c_int32), ("passed", C.c_int32)]


class EngineCfg(C.Structure):  # oth_engine_cfg
    _fields_ = [("max_games", C.c_int32), ("num_simulations", C.c_int32),
                ("temperature_threshold", C.c_int32), ("c_puct", C.c_float),
                ("dirichlet_alpha", C.c_double), ("dirichlet_epsilon", C.c_double),
                ("store_late_onehot", C.c_int32), ("eval_cache_log2", C.c_int32), ("board_size", C.c_int32)]


u64p, f32p, i32p, f64p, i64p = (C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                C.POINTER(C.c_double), C.POINTER(C.c_int64))
vp = C.c_void_p

# name -> (restype, argtypes); int-returning functions are status codes unless listed in _PLAIN_INT
_SIGS = {
    "oth_last_error": (C.c_char_p, []),
    "oth_device_available": (C.c_int, []),
    "oth_version": (C.c_char_p, []),
    "oth_board_reset": (None, [C.POINTER(Board)]),
    "oth_legal_moves": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "oth_flip_bits": (C.c_uint64, [C.c_int, C.c_uint64, C.c_uint64]),
    "oth_board_make_move": (C.c_int, [C.POINTER(Board), C.c_int]),
    "oth_board_is_terminal": (C.c_int, [C.POINTER(Board)]),
    "oth_board_get_winner": (C.c_int, [C.POINTER(Board)]),
    "oth_board_get_tensor_input": (None, [C.POINTER(Board), f32p]),
    "oth_board_get_symmetries": (None, [C.POINTER(Board), f32p, f32p, f32p]),
    "oth_legal_moves_batch": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "oth_make_move_batch": (C.c_int, [vp, vp, vp, vp, vp, C.c_int64, vp]),
    "oth_status_batch": (C.c_int, [vp, vp, vp, vp, C.c_int64, vp]),
    "oth_tensor_input_batch": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "oth_rules_checksum": (C.c_int, [C.c_int64, u64p, u64p, vp]),
    "oth_board_reset_n": (None, [C.c_int, C.POINTER(Board)]),
    "oth_legal_moves_n": (C.c_uint64, [C.c_int, C.c_uint64, C.c_uint64]),
    "oth_flip_bits_n": (C.c_uint64, [C.c_int, C.c_int, C.c_uint64, C.c_uint64]),
    "oth_board_make_move_n": (C.c_int, [C.c_int, C.POINTER(Board), C.c_int]),
    "oth_board_is_terminal_n": (C.c_int, [C.c_int, C.POINTER(Board)]),
    "oth_legal_moves_batch_n": (C.c_int, [C.c_int, vp, vp, vp, C.c_int64, vp]),
    "oth_make_move_batch_n": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int64, vp]),
    "oth_status_batch_n": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int64, vp]),
    "oth_tensor_input_batch_n": (C.c_int, [C.c_int, vp, vp, vp, C.c_int64, vp]),
    "oth_rules_checksum_n": (C.c_int, [C.c_int, C.c_int64, u64p, u64p, vp]),
    "oth_net_create": (vp, [C.c_int, C.c_int, C.c_int]),
    "oth_net_destroy": (None, [vp]),
    "oth_net_state_floats": (C.c_int64, [vp]),
    "oth_net_policy_size": (C.c_int, [vp]),
    "oth_net_load_state": (C.c_int, [vp, f32p, C.c_int64, C.c_int]),
    "oth_net_forward_bits": (C.c_int, [vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp]),
    "oth_net_forward_planes": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp]),
    "oth_net_saturated": (C.c_int, [vp, i32p, vp]),
    "oth_net_kernel_info": (C.c_int, [vp, C.c_int64, C.c_char_p, C.c_int32, f64p, f64p]),
    "oth_net_set_act_scale": (C.c_int, [vp, C.c_float]),
    "oth_net_get_act_scale": (C.c_int, [vp, f32p]),
    "oth_policy_exp": (C.c_int, [vp, vp, C.c_int64, vp]),
    "oth_engine_create": (vp, [C.POINTER(EngineCfg)]),
    "oth_engine_destroy": (None, [vp]),
    "oth_engine_set_net": (C.c_int, [vp, vp]),
    "oth_search_begin": (C.c_int, [vp, u64p, u64p, C.c_int32, vp]),
    "oth_search_select": (C.c_int, [vp, vp]),
    "oth_search_leaves": (C.c_int, [vp, i32p, u64p, u64p, u64p, vp]),
    "oth_search_expand": (C.c_int, [vp, vp, vp, C.c_int32, vp]),
    "oth_search_run": (C.c_int, [vp, vp]),
    "oth_search_results": (C.c_int, [vp, C.c_double, f32p, i32p, f64p, f32p, vp]),
    "oth_selfplay_run": (C.c_int, [vp, C.c_int32, C.c_uint64, C.c_int32, i64p, vp]),
    "oth_selfplay_begin": (C.c_int, [vp, C.c_int32, vp]),
    "oth_selfplay_search": (C.c_int, [vp, f32p, i32p, vp]),
    "oth_selfplay_apply": (C.c_int, [vp, i32p, i32p, vp]),
    "oth_selfplay_end": (C.c_int, [vp, i64p, vp]),
    "oth_stream_begin": (C.c_int, [vp, C.c_uint64, C.c_int32, C.c_int32, vp]),
    "oth_stream_step": (C.c_int, [vp, C.c_int32, i32p, i64p, vp]),
    "oth_selfplay_game_ids": (C.c_int, [vp, i32p, C.c_int32, i32p]),
    "oth_selfplay_fetch": (C.c_int, [vp, vp, vp, vp, vp, vp]),
    "oth_selfplay_device_ptrs": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64p]),
    "oth_engine_snapshot": (C.c_int, [vp, vp]),
    "oth_engine_restore": (C.c_int, [vp, vp]),
    "oth_engine_counters": (C.c_int, [vp, i64p]),
    "oth_engine_cache_stats": (C.c_int, [vp, i64p, vp]),
    "oth_engine_kernel_time": (C.c_int, [vp, f64p, i64p, f64p, i64p]),
    "oth_engine_set_timing": (C.c_int, [vp, C.c_int32]),
    "oth_engine_net_spans": (C.c_int, [vp, f64p, C.c_int64, i64p]),
    "oth_replay_gather": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp]),
    "oth_augment_symmetries": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, vp]),
    "oth_replay_gather_n": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp]),
    "oth_augment_symmetries_n": (C.c_int, [C.c_int, vp, vp, vp, C.c_int64, vp, vp, vp, vp]),
}
_PLAIN_INT = {"oth_device_available", "oth_net_policy_size", "oth_board_make_move_n", "oth_board_is_terminal_n", "oth_board_make_move", "oth_board_is_terminal", "oth_board_get_winner"}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype of the header."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python __graft_entry__.py` or `make -C %s` "
                "(hipcc --offload-arch=gfx950).  There is no pure-Python or CPU fallback."
                % (LIB_PATH, os.path.join(_HERE, "csrc")))
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code, what=""):
    if code != 0:
        msg = load().oth_last_error().decode("utf-8", "replace")
        raise OthelloHipError("%s failed (%d): %s" % (what or "libothello_mi355x call", code, msg))


def call(name, *args):
    """Call a status-returning entry point and raise OthelloHipError on failure."""
    check(getattr(load(), name)(*args), name)


def last_error():
    return load().oth_last_error().decode("utf-8", "replace")


def device_available():
    return bool(load().oth_device_available())


def require_device():
    if not device_available():
        raise OthelloHipError("no gfx950 (MI355X) device is available; the self-play engine is "
                              "HIP-only and has no CPU fallback")


def current_stream():
    """hipStream_t of torch's current stream (so the engine's work is ordered with the caller's
    torch work), or the default stream when torch.cuda is not initialised."""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    return C.c_void_p(0)


def np_ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def state_dict_blob(state_dict):
    """model.state_dict() -> the float32 blob oth_net_load_state takes: every floating tensor in
    registration order, int64 ``num_batches_tracked`` entries skipped."""
    parts = []
    for k, v in state_dict.items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        parts.append(np.asarray(a, dtype=np.float32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))
