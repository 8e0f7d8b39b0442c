"""MCTS -- mirror of the reference's single-position search object
(/root/reference/src/mcts/mcts.py:17-362), backed by the HIP engine.

Same constructor and methods; ``model`` is the trainer's nn.Module and ``device`` is accepted for
signature parity (the search always runs on the MI355X; without one every call raises).

Reference behaviours kept on purpose: the root is never backed up, so the returned
``root_value`` is always 0.0 and Dirichlet noise cannot change the visit distribution; with
``add_dirichlet_noise=True`` the noise vector is still drawn from numpy's global RNG, as the
reference does (mcts.py:221), so a seeded run consumes the same random stream.
"""
import numpy as np

from .engine import HipResNetEvaluator, SearchEngine


def best_action_from_policy(policy, legal):
    """mcts.py:288-296: the legal action with the largest probability, first one on ties."""
    best = legal[0]
    for a in legal:
        if policy[a] > policy[best]:
            best = a
    return int(best)


def evaluations_from_stats(visits, value_sum, legal):
    """mcts.py:340-360: int((Q + 1) * 50) clipped to [0, 100] for every legal action, int32 (65,)."""
    out = np.zeros(65, dtype=np.int32)
    for a in legal:
        n = int(visits[a])
        q = 0.0 if n == 0 else float(value_sum[a]) / n
        out[a] = max(0, min(100, int((q + 1.0) * 50.0)))
    return out


class MCTS:
    def __init__(self, model, device=None, c_puct=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                 precision=None):
        self.model = model
        self.device = device
        self.c_puct = c_puct
        self.dirichlet_alpha = dirichlet_alpha
        self.dirichlet_epsilon = dirichlet_epsilon
        self.evaluator = HipResNetEvaluator(model, precision=precision)
        self._engines = {}

    def _engine(self, num_simulations, max_games=1):
        key = (int(num_simulations), int(max_games))
        eng = self._engines.get(key)
        if eng is None:
            eng = SearchEngine(max_games, num_simulations, c_puct=self.c_puct,
                               dirichlet_alpha=self.dirichlet_alpha,
                               dirichlet_epsilon=self.dirichlet_epsilon, evaluator=self.evaluator)
            self._engines[key] = eng
        return eng

    def _search_stats(self, board, num_simulations, temperature):
        self.evaluator.refresh()
        eng = self._engine(num_simulations)
        eng.search_begin([board.self_board], [board.opp_board])
        eng.search_run_rescued()   # (a launch that clamped an activation is run again at a lower activation scale)
        return eng.search_results(temperature)

    def search(self, board, num_simulations, temperature=1.0, add_dirichlet_noise=False):
        """-> (policy (65,) float32, root_value).  mcts.py:49-98."""
        if add_dirichlet_noise:  # mcts.py:85-86 / :220-221: consumes the RNG, cannot alter the result
            n_legal = len(board.get_legal_moves())
            np.random.dirichlet([self.dirichlet_alpha] * n_legal)
        pi, _, _, _ = self._search_stats(board, num_simulations, float(temperature))
        return pi[0].copy(), 0.0

    def get_action_probs(self, board, num_simulations, temperature=1.0, add_dirichlet_noise=False):
        policy, _ = self.search(board, num_simulations, temperature, add_dirichlet_noise)  # mcts.py:230
        return policy

    def get_best_action(self, board, num_simulations):
        """mcts.py:257-296"""
        legal = board.get_legal_moves()
        if num_simulations < 1:
            return legal[0]
        policy, _ = self.search(board, num_simulations, temperature=0.0, add_dirichlet_noise=False)
        return best_action_from_policy(policy, legal)

    def get_action_evaluations(self, board, num_simulations):
        """mcts.py:298-362: int((Q+1)*50) clipped to [0,100] per legal action, int32 (65,)."""
        legal = board.get_legal_moves()
        if num_simulations < 1:
            return np.zeros(65, dtype=np.int32)
        _, visits, wsum, _ = self._search_stats(board, num_simulations, 1.0)
        return evaluations_from_stats(visits[0], wsum[0], legal)
