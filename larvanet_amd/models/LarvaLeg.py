"""LarvaNet with early-exit inference: drop-in for the reference plugin models/LarvaLeg.py.

Same network and checkpoints as LarvaNet; `--leg=k` makes forward() stop after body k-1 and
return that body's exit (k = num_modules is the full network, k = 0 returns the bicubic base
image alone) -- models/LarvaLeg.py:52, 275, 289-300.  Training is unchanged (all exits)."""
import torch

from ..autograd import DualChain
from . import LarvaNet as V1
from .LarvaNetV2 import LarvaNet as _V2Wrapper


def create_model():
    return LarvaNet()


class LarvaNetModule(V1.LarvaNetModule):
    def __init__(self, args):
        super().__init__(args)
        self.leg = args.leg
        if not 0 <= self.leg <= self.len:
            raise ValueError("--leg must be in [0, num_modules]")

    def forward(self, x):
        base = self.base(x)
        if self.leg == 0:
            return base
        with self.width_scope(x):
            fea = self.head(x)
            for i in range(self.leg):
                fea = getattr(self, "body_%d" % i)(fea)
            DualChain.join()
            return getattr(self, "body_%d" % (self.leg - 1)).leg(fea, base)


class LarvaNet(V1.LarvaNet):
    module_class = LarvaNetModule

    def _add_args(self, parser):
        # flag set and defaults of models/LarvaLeg.py:46-61 (= LarvaNetV2's plus --leg)
        _V2Wrapper._add_args(self, parser)
        parser.add_argument("--leg", type=int, default=4, help="The early exit leg number, starts at 1.")

    def receptive_halo(self):
        k = self.args.leg
        return 2 if k == 0 else 1 + 2 * sum(V1.parse_num_blocks(self.args)[:k]) + 2

    def _make_scheduler(self):
        return torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optim, mode="max", factor=self.args.lr_decay, patience=self.args.patience,
            threshold=self.args.threshold, threshold_mode="abs", min_lr=self.args.min_lr)
