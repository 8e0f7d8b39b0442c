"""MI355X-native self-play engine behind the reference's Bitboard / MCTS / SelfPlayWorker API.

Drop-in surface (reference module -> here):
  src.cython.bitboard.OthelloBitboard            -> bitboard.OthelloBitboard
  src.mcts.mcts.MCTS                             -> mcts.MCTS
  src.mcts.node.MCTSNode                         -> node.MCTSNode (host object view; the engine's trees are device arrays)
  src.train.self_play.SelfPlayWorker             -> self_play.SelfPlayWorker
  src.train.parallel_self_play.BatchMCTS         -> parallel_self_play.BatchMCTS
  src.train.parallel_self_play.ParallelSelfPlayWorker / create_parallel_self_play_worker
                                                 -> parallel_self_play.*
  src.model.net.OthelloResNet (trainer side)     -> net.OthelloResNet (same state_dict)
  src.eval.players / src.eval.arena              -> arena.* (+ BatchedArena on the device search)
  src.train.buffer.ReplayBuffer                  -> replay.DeviceReplayBuffer (device-resident)

Everything that computes runs in libothello_mi355x.so (hand-written HIP for gfx950, built in-tree);
importing the compute classes without that library raises ImportError, and using them without an
MI355X raises OthelloHipError.  There is no CPU fallback.
"""
from ._lib import OthelloHipError, device_available  # noqa: F401
from .arena import (Arena, BatchedArena, GreedyPlayer, MatchResult, MCTSPlayer, Player,  # noqa: F401
                    RandomPlayer, evaluate_player)
from .bitboard import DeviceBoards, OthelloBitboard  # noqa: F401
from .engine import HipResNetEvaluator, SearchEngine  # noqa: F401
from .mcts import MCTS  # noqa: F401
from .node import MCTSNode  # noqa: F401
from .net import OthelloResNet, create_model  # noqa: F401
from .parallel_self_play import (BatchMCTS, ParallelSelfPlayWorker,  # noqa: F401
                                 create_parallel_self_play_worker)
from .replay import (DeviceReplayBuffer, augment_symmetries, augment_training_data,  # noqa: F401
                     load_checkpoint_model)
from .self_play import GameStep, SelfPlayWorker, augment_data_with_symmetries  # noqa: F401

__all__ = [
    "OthelloBitboard", "DeviceBoards", "MCTS", "MCTSNode", "BatchMCTS", "SelfPlayWorker", "ParallelSelfPlayWorker",
    "create_parallel_self_play_worker", "GameStep", "augment_data_with_symmetries", "OthelloResNet",
    "create_model", "HipResNetEvaluator", "SearchEngine", "OthelloHipError", "device_available",
    "DeviceReplayBuffer", "augment_symmetries", "augment_training_data", "load_checkpoint_model",
    "Arena", "BatchedArena", "Player", "RandomPlayer", "GreedyPlayer", "MCTSPlayer", "MatchResult", "evaluate_player",
]
