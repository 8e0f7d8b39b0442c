"""OthelloBitboard -- Python mirror of the reference's Cython class
(/root/reference/src/cython/bitboard.pyx:41, field/method list at bitboard.pxd:25-48).

A single board object is a host-side value (two uint64 + two ints); its methods call the host
entry points of libothello_mi355x.so, which are compiled from the SAME ``othello_rules.h`` source
as the device kernels.  Bulk work (self-play, search) never goes through this class: it runs in
the HIP engine (engine.py) on arrays of positions.  ``DeviceBoards`` below is the array form of
the same methods, running on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib


class OthelloBitboard:
    """Same public surface as the reference class: fields ``self_board``, ``opp_board``,
    ``move_count``, ``passed`` (readable and writable) and the 11 methods of bitboard.pxd:38-48."""

    __slots__ = ("_b",)

    def __init__(self):
        self._b = _lib.Board()
        _lib.load().oth_board_reset(C.byref(self._b))

    # ---- fields (bitboard.pxd:25-28) -------------------------------------------------------
    @property
    def self_board(self):
        return int(self._b.self_board)

    @self_board.setter
    def self_board(self, v):
        self._b.self_board = int(v) & 0xFFFFFFFFFFFFFFFF

    @property
    def opp_board(self):
        return int(self._b.opp_board)

    @opp_board.setter
    def opp_board(self, v):
        self._b.opp_board = int(v) & 0xFFFFFFFFFFFFFFFF

    @property
    def move_count(self):
        return int(self._b.move_count)

    @move_count.setter
    def move_count(self, v):
        self._b.move_count = int(v)

    @property
    def passed(self):
        return bool(self._b.passed)

    @passed.setter
    def passed(self, v):
        self._b.passed = 1 if v else 0

    # ---- methods ---------------------------------------------------------------------------
    def reset(self):  # bitboard.pyx:52
        _lib.load().oth_board_reset(C.byref(self._b))

    def get_legal_moves_bits(self):  # bitboard.pyx:187
        return int(_lib.load().oth_legal_moves(self._b.self_board, self._b.opp_board))

    def get_legal_moves(self):  # bitboard.pyx:166: ascending list, [64] when there is no move
        bits = self.get_legal_moves_bits()
        if bits == 0:
            return [64]
        return [i for i in range(64) if (bits >> i) & 1]

    def make_move(self, pos):  # bitboard.pyx:195: False (state unchanged) for an invalid move
        return bool(_lib.load().oth_board_make_move(C.byref(self._b), int(pos)))

    def is_terminal(self):  # bitboard.pyx:249
        return bool(_lib.load().oth_board_is_terminal(C.byref(self._b)))

    def get_winner(self):  # bitboard.pyx:266: relative to the side to move
        return int(_lib.load().oth_board_get_winner(C.byref(self._b)))

    def get_stone_counts(self):  # bitboard.pyx:292
        return (bin(self.self_board).count("1"), bin(self.opp_board).count("1"))

    def get_tensor_input(self):  # bitboard.pyx:300: float32 (3,8,8): own, opponent, legal planes
        t = np.empty((3, 8, 8), dtype=np.float32)
        _lib.load().oth_board_get_tensor_input(C.byref(self._b), _lib.np_ptr(t, C.c_float))
        return t

    def copy(self):  # bitboard.pyx:325
        new = OthelloBitboard.__new__(OthelloBitboard)
        new._b = _lib.Board(self._b.self_board, self._b.opp_board, self._b.move_count, self._b.passed)
        return new

    def get_symmetries(self, pi):  # bitboard.pyx:338: 8 (state, pi) pairs
        pi = np.ascontiguousarray(pi, dtype=np.float32)
        st = np.empty((8, 3, 8, 8), dtype=np.float32)
        ps = np.empty((8, 65), dtype=np.float32)
        _lib.load().oth_board_get_symmetries(C.byref(self._b), _lib.np_ptr(pi, C.c_float),
                                             _lib.np_ptr(st, C.c_float), _lib.np_ptr(ps, C.c_float))
        return [(st[k].copy(), ps[k].copy()) for k in range(8)]

    def to_string(self):  # bitboard.pyx:392
        return repr(self)

    def __repr__(self):  # bitboard.pyx:372: same glyphs and layout
        lines = ["  A B C D E F G H"]
        s, o = self.self_board, self.opp_board
        for row in range(8):
            line = "%d " % (row + 1)
            for col in range(8):
                i = row * 8 + col
                line += "● " if (s >> i) & 1 else ("○ " if (o >> i) & 1 else ". ")
            lines.append(line)
        return "\n".join(lines)


class DeviceBoards:
    """The same rules over arrays of positions on the GPU (torch int64 tensors holding the uint64
    bit patterns).  Thin wrappers over section 2 of the C ABI.  ``board_size=6`` selects the 6x6 rules (bit i =
    row*6+col, pass 36, planes [3,6,6]; parity unpinned -- the reference has no 6x6 game)."""

    @staticmethod
    def _args(*tensors):
        import torch
        _lib.require_device()
        for t in tensors:
            if t is not None and (not t.is_cuda or not t.is_contiguous()):
                raise ValueError("DeviceBoards expects contiguous CUDA tensors")
        return torch

    @staticmethod
    def legal_moves(self_b, opp_b, board_size=8):
        torch = DeviceBoards._args(self_b, opp_b)
        out = torch.empty_like(self_b)
        _lib.call("oth_legal_moves_batch_n", int(board_size), self_b.data_ptr(), opp_b.data_ptr(), out.data_ptr(),
                  self_b.numel(), _lib.current_stream())
        return out

    @staticmethod
    def make_move(self_b, opp_b, pos, board_size=8):
        """In place.  Returns (ok int32[n], flips int64[n])."""
        torch = DeviceBoards._args(self_b, opp_b, pos)
        ok = torch.empty(self_b.numel(), dtype=torch.int32, device=self_b.device)
        flips = torch.empty_like(self_b)
        _lib.call("oth_make_move_batch_n", int(board_size), self_b.data_ptr(), opp_b.data_ptr(), pos.data_ptr(),
                  ok.data_ptr(), flips.data_ptr(), self_b.numel(), _lib.current_stream())
        return ok, flips

    @staticmethod
    def status(self_b, opp_b, board_size=8):
        """-> (terminal int32[n], winner int32[n])"""
        torch = DeviceBoards._args(self_b, opp_b)
        term = torch.empty(self_b.numel(), dtype=torch.int32, device=self_b.device)
        win = torch.empty_like(term)
        _lib.call("oth_status_batch_n", int(board_size), self_b.data_ptr(), opp_b.data_ptr(), term.data_ptr(),
                  win.data_ptr(), self_b.numel(), _lib.current_stream())
        return term, win

    @staticmethod
    def tensor_input(self_b, opp_b, board_size=8):
        torch = DeviceBoards._args(self_b, opp_b)
        bs = int(board_size)
        out = torch.empty((self_b.numel(), 3, bs, bs), dtype=torch.float32, device=self_b.device)
        _lib.call("oth_tensor_input_batch_n", bs, self_b.data_ptr(), opp_b.data_ptr(), out.data_ptr(),
                  self_b.numel(), _lib.current_stream())
        return out
