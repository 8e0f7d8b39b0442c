"""Players and Arena -- mirrors of /root/reference/src/eval/players.py and arena.py.

The host-side players (random, greedy) and the match bookkeeping restate the reference literally,
including how it derives the result: ``board.get_winner()`` / ``get_stone_counts()`` are relative to the
side to move at the end of the game and the reference maps them as if they were black/white
(arena.py:126-151), and ``GreedyPlayer`` picks the count by move-count parity (players.py:93-101).
Golden vectors produced by the reference's own Arena pin this (tests/golden/g6_arena.npz).

``MCTSPlayer`` runs its searches on the HIP engine.  ``BatchedArena`` plays N matches in lock-step: all
boards where the searching player is to move go through ONE batched device search per ply
(``BatchMCTS.search_batch`` at temperature 0 == ``get_best_action``), which is what makes model evaluation
throughput-bound by the same kernels as self-play.
"""
import random
import time
from dataclasses import dataclass
from typing import List

from .bitboard import OthelloBitboard
from .mcts import MCTS, best_action_from_policy


class Player:
    """players.py:20-48"""

    def __init__(self, name):
        self.name = name

    def get_action(self, board):
        raise NotImplementedError

    def reset(self):
        pass


class RandomPlayer(Player):
    """players.py:50-67: ``random.choice`` over the legal moves (python's global RNG, like the reference)."""

    def __init__(self, name="Random"):
        super().__init__(name)

    def get_action(self, board):
        legal = board.get_legal_moves()
        if len(legal) == 0:
            return 64
        return random.choice(legal)


class GreedyPlayer(Player):
    """players.py:70-112: the move after which the mover's stone count (as the reference reads it) is largest;
    first such move on ties."""

    def __init__(self, name="Greedy"):
        super().__init__(name)

    def get_action(self, board):
        legal = board.get_legal_moves()
        if len(legal) == 0:
            return 64
        best_action, best_score = legal[0], -1
        for action in legal:
            test = board.copy()
            test.make_move(action)
            first, second = test.get_stone_counts()          # (side to move, other) AFTER the move
            score = second if board.move_count % 2 == 0 else first   # players.py:93-101, literally
            if score > best_score:
                best_score, best_action = score, action
        return best_action


class MCTSPlayer(Player):
    """players.py:115-157 on the HIP search (c_puct 1.0, temperature 0)."""

    def __init__(self, model, device=None, num_simulations=50, name="MCTS-AI", precision=None):
        super().__init__(name)
        self.model = model
        self.device = device
        self.num_simulations = num_simulations
        self.mcts = MCTS(model=model, device=device, c_puct=1.0, precision=precision)

    def get_action(self, board):
        return self.mcts.get_best_action(board, num_simulations=self.num_simulations)

    @classmethod
    def from_checkpoint(cls, checkpoint_path, device=None, num_simulations=50):
        """players.py:159-211 (architecture inferred from the key names; loaded with weights_only=True)."""
        from .replay import load_checkpoint_model
        model = load_checkpoint_model(checkpoint_path)
        return cls(model=model, device=device, num_simulations=num_simulations,
                   name="MCTS-AI-%dsim" % num_simulations)


@dataclass
class MatchResult:
    """arena.py:13-52"""
    player1_name: str
    player2_name: str
    winner: int            # 1: player1, -1: player2, 0: draw
    player1_score: int
    player2_score: int
    num_moves: int
    duration: float

    def __str__(self):
        if self.winner == 1:
            res = "%s wins" % self.player1_name
        elif self.winner == -1:
            res = "%s wins" % self.player2_name
        else:
            res = "Draw"
        return "%s | %s: %d - %s: %d | Moves: %d | Time: %.2fs" % (
            res, self.player1_name, self.player1_score, self.player2_name, self.player2_score,
            self.num_moves, self.duration)


def _result_from_final_board(board, p1_name, p2_name, starting_player, duration):
    """arena.py:122-162, literally: get_winner()/get_stone_counts() read as (black, white)."""
    winner_color = board.get_winner()
    black_count, white_count = board.get_stone_counts()
    if starting_player == 1:
        winner = 1 if winner_color == 1 else (-1 if winner_color == -1 else 0)
        s1, s2 = black_count, white_count
    else:
        winner = -1 if winner_color == 1 else (1 if winner_color == -1 else 0)
        s1, s2 = white_count, black_count
    return MatchResult(p1_name, p2_name, winner, s1, s2, board.move_count, duration)


class Arena:
    """arena.py:55-232"""

    def __init__(self, verbose=True):
        self.verbose = verbose

    def play_game(self, player1, player2, starting_player=1):
        board = OthelloBitboard()
        board.reset()
        player1.reset()
        player2.reset()
        current, other = (player1, player2) if starting_player == 1 else (player2, player1)
        t0 = time.time()
        while not board.is_terminal():
            action = current.get_action(board)
            if self.verbose:
                print("%s plays: %s (legal: %s)" % (current.name, action, board.get_legal_moves()))
            board.make_move(action)
            current, other = other, current
        result = _result_from_final_board(board, player1.name, player2.name, starting_player, time.time() - t0)
        if self.verbose:
            print("\n%s\n" % result)
        return result

    def play_matches(self, player1, player2, num_games=10, alternate_colors=True) -> List[MatchResult]:
        results = []
        for g in range(num_games):
            if self.verbose:
                print("=== Game %d/%d ===" % (g + 1, num_games))
            start = (1 if g % 2 == 0 else -1) if alternate_colors else 1
            results.append(self.play_game(player1, player2, start))
        if self.verbose:
            self._print_summary(results, player1.name, player2.name)
        return results

    def _print_summary(self, results, n1, n2):
        total = len(results)
        w1 = sum(1 for r in results if r.winner == 1)
        w2 = sum(1 for r in results if r.winner == -1)
        print("\n" + "=" * 70 + "\nMatch Summary\n" + "=" * 70)
        print("\nTotal Games: %d" % total)
        print("%s: %d wins (%.1f%%)" % (n1, w1, w1 / total * 100 if total else 0))
        print("%s: %d wins (%.1f%%)" % (n2, w2, w2 / total * 100 if total else 0))
        print("Draws: %d" % (total - w1 - w2))
        print("\nAverage Moves: %.1f" % (sum(r.num_moves for r in results) / total if total else 0))
        print("Average Duration: %.2fs" % (sum(r.duration for r in results) / total if total else 0))
        print("=" * 70 + "\n")


def summarize(results, num_games):
    """The dict evaluate_player returns (arena.py:262-275)."""
    wins = sum(1 for r in results if r.winner == 1)
    return {
        "win_rate": wins / num_games if num_games > 0 else 0,
        "avg_score": sum(r.player1_score for r in results) / num_games if num_games > 0 else 0,
        "avg_moves": sum(r.num_moves for r in results) / num_games if num_games > 0 else 0,
        "results": results,
    }


def evaluate_player(player, opponent, num_games=10, verbose=True):
    """arena.py:235-275"""
    return summarize(Arena(verbose=verbose).play_matches(player, opponent, num_games=num_games), num_games)


class BatchedArena:
    """N matches of an ``MCTSPlayer`` (player1) against a host-side opponent in lock-step.  Per ply, every board
    where the MCTS player is to move is searched in ONE batched device call; the opponent's moves are computed
    on the host game by game, in game order (so a seeded RandomPlayer consumes python's RNG deterministically).
    For deterministic opponents the results equal ``Arena.play_matches`` game for game."""

    def __init__(self, batch_mcts, num_simulations=50):
        self.batch_mcts = batch_mcts          # parallel_self_play.BatchMCTS
        self.num_simulations = num_simulations

    def play_matches(self, player1_name, opponent, num_games=10, alternate_colors=True):
        boards = [OthelloBitboard() for _ in range(num_games)]
        starts = [(1 if g % 2 == 0 else -1) if alternate_colors else 1 for g in range(num_games)]
        t0 = time.time()
        done = [False] * num_games
        while not all(done):
            # whose move: player1 moves on even plies iff it started
            mine = [g for g in range(num_games) if not done[g] and
                    ((boards[g].move_count % 2 == 0) == (starts[g] == 1))]
            theirs = [g for g in range(num_games) if not done[g] and g not in set(mine)]
            if mine:
                res = self.batch_mcts.search_batch([boards[g] for g in mine], self.num_simulations,
                                                   temperature=0.0, add_dirichlet_noise=False)
                for g, (pi, _) in zip(mine, res):
                    boards[g].make_move(best_action_from_policy(pi, boards[g].get_legal_moves()))
            for g in theirs:
                boards[g].make_move(opponent.get_action(boards[g]))
            for g in range(num_games):
                if not done[g] and boards[g].is_terminal():
                    done[g] = True
        dt = (time.time() - t0) / max(1, num_games)
        return [_result_from_final_board(boards[g], player1_name, opponent.name, starts[g], dt)
                for g in range(num_games)]
