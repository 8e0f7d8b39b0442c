"""Policy/value network definition (PyTorch side) for the MI355X self-play engine.

The training loop stays in PyTorch (it is the *caller* of the hot path),
so this module keeps an ``nn.Module`` with exactly the parameter names, registration order and
forward semantics of the reference's ``OthelloResNet`` (/root/reference/src/model/net.py:139-205):
identical ``state_dict`` keys (144 entries for 10x128) and, under the same torch seed, identical
initial weights.  Checkpoints are therefore interchangeable in both directions.

Self-play inference does NOT run through this module: ``HipResNetEvaluator`` (evaluator.py) folds
the BatchNorms, packs the weights into MFMA fragment order and runs the hand-written gfx950
kernel.  ``forward`` here is what the trainer differentiates and what the parity tests compare the
kernel against (tolerance 1e-4, BASELINE.json north_star).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBlock(nn.Module):
    """3x3 conv (no bias) -> BN -> ReLU   (reference net.py:15-31)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class ResBlock(nn.Module):
    """conv-bn-relu-conv-bn, add skip, relu   (reference net.py:34-61)."""

    def __init__(self, num_filters):
        super().__init__()
        self.conv1 = nn.Conv2d(num_filters, num_filters, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(num_filters)
        self.conv2 = nn.Conv2d(num_filters, num_filters, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(num_filters)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class PolicyHead(nn.Module):
    """1x1 conv to 2 planes -> BN -> ReLU -> FC(2*S*S -> S*S+1) -> log_softmax (net.py:64-96)."""

    def __init__(self, num_filters, board_size=8):
        super().__init__()
        self.board_size = board_size
        self.conv = nn.Conv2d(num_filters, 2, 1, bias=False)
        self.bn = nn.BatchNorm2d(2)
        self.fc = nn.Linear(2 * board_size * board_size, board_size * board_size + 1)

    def forward(self, x):
        x = F.relu(self.bn(self.conv(x)))
        return F.log_softmax(self.fc(x.flatten(1)), dim=1)


class ValueHead(nn.Module):
    """1x1 conv to 1 plane -> BN -> ReLU -> FC(S*S -> 256) -> ReLU -> FC(256 -> 1) -> tanh
    (reference net.py:99-136)."""

    def __init__(self, num_filters, board_size=8, hidden_size=256):
        super().__init__()
        self.board_size = board_size
        self.conv = nn.Conv2d(num_filters, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(1)
        self.fc1 = nn.Linear(board_size * board_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, 1)

    def forward(self, x):
        x = F.relu(self.bn(self.conv(x)))
        x = F.relu(self.fc1(x.flatten(1)))
        return torch.tanh(self.fc2(x))


class OthelloResNet(nn.Module):
    """Dual-head ResNet: input (N,3,8,8) = own / opponent / legal-move planes,
    outputs (log-probabilities (N,65), value (N,1))."""

    def __init__(self, num_blocks=10, num_filters=128, board_size=8):
        super().__init__()
        self.num_blocks = num_blocks
        self.num_filters = num_filters
        self.board_size = board_size
        self.conv_block = ConvBlock(3, num_filters)
        self.res_blocks = nn.ModuleList([ResBlock(num_filters) for _ in range(num_blocks)])
        self.policy_head = PolicyHead(num_filters, board_size)
        self.value_head = ValueHead(num_filters, board_size)

    def forward(self, x):
        x = self.conv_block(x)
        for blk in self.res_blocks:
            x = blk(x)
        return self.policy_head(x), self.value_head(x)

    def predict(self, board_tensor):
        """Probabilities (not log) and value, accepting a single (3,8,8) position too
        (reference net.py:207-236)."""
        single = board_tensor.dim() == 3
        if single:
            board_tensor = board_tensor.unsqueeze(0)
        self.eval()
        with torch.no_grad():
            logp, v = self.forward(board_tensor)
            p = torch.exp(logp)
        if single:
            p, v = p.squeeze(0), v.squeeze(0)
        return p, v

    def get_param_count(self):
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return {"total": total, "trainable": trainable}


def create_model(config):
    """Build from the ``model:`` section of a reference YAML config (net.py:244-265)."""
    return OthelloResNet(
        num_blocks=config.get("num_blocks", 10),
        num_filters=config.get("num_filters", 128),
        board_size=config.get("board_size", 8),
    )
