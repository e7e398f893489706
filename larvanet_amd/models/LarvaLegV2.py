"""LarvaNetV2 with early-exit inference: drop-in for the reference plugin models/LarvaLegV2.py.

The V2 network and its checkpoints (tail.* keys included; training is V2's: every exit plus the tail,
models/LarvaLegV2.py:102-124), but forward() stops after body k-1 and returns that body's exit for `--leg=k`
(k = 0: the bicubic base image alone) -- the tail is not evaluated (models/LarvaLegV2.py:52, 342, 358-367)."""
from ..autograd import DualChain
from . import LarvaNet as V1
from . import LarvaNetV2 as V2


def create_model():
    return LarvaNet()


class LarvaNetModule(V2.LarvaNetModule):
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


class LarvaNet(V2.LarvaNet):
    module_class = LarvaNetModule

    def _add_args(self, parser):
        # flag set and defaults of models/LarvaLegV2.py:46-61 (= LarvaNetV2's plus --leg)
        V2.LarvaNet._add_args(self, parser)
        parser.add_argument("--leg", type=int, default=4, help="The early exit leg number, starts at 1.")

    def receptive_halo(self):
        k = self.args.leg
        return 2 if k == 0 else 1 + 2 * sum(V1.parse_num_blocks(self.args)[:k]) + 2
