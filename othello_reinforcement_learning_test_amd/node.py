"""MCTSNode -- host-side mirror of /root/reference/src/mcts/node.py:12-190.

The engine keeps trees as node/edge arrays on the device (csrc/engine.hip); this class exists for callers that
build or inspect a tree by hand with the reference's object API (the reference's own tests do).  The arithmetic is
the reference's, operation for operation (numpy float32 priors, float64 PUCT with the float32 `c_puct * prior`
product of NumPy 2, first-maximum ties) -- the same arithmetic the device kernels reproduce.
``SearchEngine.search_results`` gives the device tree's root statistics; ``from_root_stats`` turns them into a
root node with children for code that wants the object view.
"""
import numpy as np


class MCTSNode:
    def __init__(self, prior, parent=None):
        self.prior = prior
        self.parent = parent
        self.visit_count = 0      # N(s,a)
        self.value_sum = 0.0      # W(s,a)
        self.children = {}        # action -> MCTSNode, insertion order = legal-move order
        self.is_expanded = False

    def is_leaf(self):
        return len(self.children) == 0

    def get_value(self):  # node.py:51-60
        return 0.0 if self.visit_count == 0 else self.value_sum / self.visit_count

    def expand(self, policy_probs, legal_actions):  # node.py:62-89
        masked = np.zeros_like(policy_probs)
        masked[legal_actions] = policy_probs[legal_actions]
        total = masked.sum()
        if total > 0:
            masked /= total
        else:
            masked[legal_actions] = 1.0 / len(legal_actions)
        for a in legal_actions:
            self.children[a] = MCTSNode(prior=masked[a], parent=self)
        self.is_expanded = True

    def select_child(self, c_puct):  # node.py:91-126
        best_score, best_action, best_child = -float("inf"), None, None
        n_parent = self.visit_count
        for action, child in self.children.items():
            u = c_puct * child.prior * np.sqrt(n_parent) / (1 + child.visit_count)
            score = child.get_value() + u
            if score > best_score:
                best_score, best_action, best_child = score, action, child
        return best_action, best_child

    def update(self, value):  # node.py:128-136
        self.visit_count += 1
        self.value_sum += value

    def get_visit_counts(self):
        return {a: c.visit_count for a, c in self.children.items()}

    def get_policy_distribution(self, temperature=1.0):  # node.py:147-182
        policy = np.zeros(65, dtype=np.float32)
        if not self.children:
            return policy
        actions = list(self.children.keys())
        counts = np.array([self.children[a].visit_count for a in actions], dtype=np.float32)
        if temperature == 0:
            policy[actions[int(np.argmax(counts))]] = 1.0
        else:
            counts = counts ** (1.0 / temperature)
            counts /= counts.sum()
            for a, p in zip(actions, counts):
                policy[a] = p
        return policy

    @classmethod
    def from_root_stats(cls, visits, value_sum, prior):
        """Root node (with one level of children) from the device statistics of one position:
        the (65,) rows of ``SearchEngine.search_results``."""
        root = cls(prior=1.0)
        for a in range(65):
            if prior[a] != 0 or visits[a] != 0:
                ch = cls(prior=np.float32(prior[a]), parent=root)
                ch.visit_count = int(visits[a])
                ch.value_sum = float(value_sum[a])
                root.children[a] = ch
        root.is_expanded = bool(root.children)
        return root

    def __repr__(self):
        return "MCTSNode(prior=%.3f, N=%d, W=%.3f, Q=%.3f, children=%d)" % (
            self.prior, self.visit_count, self.value_sum, self.get_value(), len(self.children))
