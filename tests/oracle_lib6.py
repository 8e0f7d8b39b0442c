"""The 6x6 build of the CPU oracle (oracle/libothello_oracle6.so = othello_oracle.c with -DORC_N=6), bound by
re-executing oracle_lib.py with BOARD = 6.  PARITY UNPINNED: the reference has no 6x6 rules; this checker exists so
that the engine's 6x6 kernel path (BASELINE configs[4]) is compared with an independent CPU implementation of the
same definition, bit for bit.  Test infrastructure only."""
import importlib.util
import os
import sys

_spec = importlib.util.spec_from_file_location(__name__, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                      "oracle_lib.py"))
_mod = importlib.util.module_from_spec(_spec)
_mod._BOARD_OVERRIDE = 6
_spec.loader.exec_module(_mod)
sys.modules[__name__] = _mod
