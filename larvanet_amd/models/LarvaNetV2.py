"""LarvaNetV2 for MI355X: drop-in for the reference plugin models/LarvaNetV2.py.

V1 plus a LarvaTail that merges the features of ALL bodies: torch.cat(features, 1) ->
merge_conv (48*M -> 48, no activation) -> conv/ReLU/conv -> PixelShuffle(4) -> + base
(models/LarvaNetV2.py:314-334).  Inference uses the tail only (:355-365); training adds the
tail's L1 loss to the per-exit losses and divides by M+1 (:101-123).  The concatenation is never
materialised: the conv kernel walks the M feature tensors as consecutive K chunks.

state_dict adds tail.merge_conv.{weight,bias} and tail.recon_block.{0,2}.{weight,bias} to V1's keys.
"""
import torch
import torch.nn as nn

from ..autograd import DualChain, LegFn, MergeFn, PackedConv, mean_of_terms
from . import LarvaNet as V1
from .LarvaNet import NUM_FILTERS, _conv, _require_hip, init_conv


def create_model():
    return LarvaNet()


class LarvaTail(nn.Module):
    """models/LarvaNetV2.py:314-334"""

    def __init__(self, num_modules, num_filters=NUM_FILTERS):
        super().__init__()
        self.merge_conv = _conv(num_filters * num_modules, num_filters)
        self.recon_block = nn.Sequential(_conv(num_filters, num_filters), nn.ReLU(inplace=True),
                                         _conv(num_filters, NUM_FILTERS))
        # the reference initialises [recon_block, merge_conv] in that order (:324)
        init_conv(self.recon_block[0])
        init_conv(self.recon_block[2])
        init_conv(self.merge_conv)
        self.upsample = nn.PixelShuffle(4)
        self._pc = PackedConv(self.merge_conv.weight, self.merge_conv.bias,
                              slices=[(i * num_filters, num_filters) for i in range(num_modules)])
        self._pcs = [PackedConv(self.recon_block[0].weight, self.recon_block[0].bias),
                     PackedConv(self.recon_block[2].weight, self.recon_block[2].bias)]

    def forward(self, features, base):
        _require_hip(features[0])
        self._pc.refresh()
        for pc in self._pcs:
            pc.refresh()
        m = self.merge_conv
        fea = MergeFn.apply(self._pc, m.weight, m.bias, *[f.contiguous() for f in features])
        c1, c2 = self.recon_block[0], self.recon_block[2]
        return LegFn.apply(fea, base.contiguous(), self._pcs, c1.weight, c1.bias, c2.weight, c2.bias)


class LarvaNetModule(V1.LarvaNetModule):
    """models/LarvaNetV2.py:337-365"""

    def __init__(self, args):
        super().__init__(args)
        self.tail = LarvaTail(self.len, self.num_filters)

    def features(self, x):
        fea = self.head(x)
        feats = []
        for i in range(self.len):
            fea = getattr(self, "body_%d" % i)(fea)
            feats.append(fea)
        return feats

    def forward(self, x):
        with self.width_scope(x):
            base = self.base(x)
            feats = self.features(x)
            DualChain.join()
            return self.tail(feats, base)


class LarvaNet(V1.LarvaNet):
    """Plugin wrapper; control flow of models/LarvaNetV2.py:41-212."""

    module_class = LarvaNetModule

    def _add_args(self, parser):
        # V2's flag set and defaults (models/LarvaNetV2.py:46-60): no --lr_step / --cooldown
        parser.add_argument("--num_modules", type=int, default=2)
        parser.add_argument("--num_blocks", type=str, default="16,16")
        parser.add_argument("--interpolate", type=str, default="bicubic")
        parser.add_argument("--val_volume", type=float, default=3e9)
        parser.add_argument("--lr", type=float, default=1e-4)
        parser.add_argument("--lr_decay", type=float, default=0.5)
        parser.add_argument("--threshold", type=float, default=0.001)
        parser.add_argument("--min_lr", type=float, default=1e-7)
        parser.add_argument("--patience", type=int, default=3)
        self._add_build_args(parser)

    def _make_scheduler(self):
        return torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optim, mode="max", factor=self.args.lr_decay, patience=self.args.patience,
            threshold=self.args.threshold, threshold_mode="abs", min_lr=self.args.min_lr)

    def _num_loss_terms(self):
        return self.args.num_modules + 1

    def _single_consumer_features(self):
        return False   # the tail's merge conv reads every body output: autograd adds its gradient on the main stream

    def _exit_losses(self, input_tensor, truth_tensor):
        """models/LarvaNetV2.py:104-123: every exit plus the tail, / (M + 1)."""
        net = self.model
        base, x16 = net.step_prologue(input_tensor)
        fea = net.head(input_tensor, x16)
        terms = []
        feats = []
        if self._exits_batched():
            for i in range(self.args.num_modules):
                fea = getattr(net, "body_%d" % i)(fea)
                feats.append(fea)
            DualChain.join()
            _, terms = self._all_exits(feats, base, truth_tensor)
        else:
            for i in range(self.args.num_modules):
                body = getattr(net, "body_%d" % i)
                fea = body(fea)
                feats.append(fea)
                terms.append(self._exit(body.leg, fea, base, truth_tensor)[1])
        out = net.tail(feats, base)
        terms.append(self.loss_fn(out, truth_tensor))
        self._sync_exits()
        return mean_of_terms(terms), out

    def receptive_halo(self):
        # head + bodies + merge conv + the tail's two convs
        return 1 + 2 * sum(V1.parse_num_blocks(self.args)) + 1 + 2

    def restore(self, ckpt_path, target=None):
        """Only keys present in this network are taken (V1 checkpoints warm-start V2),
        models/LarvaNetV2.py:196-206."""
        pretrained = torch.load(ckpt_path, map_location=self.device)
        current = self.model.state_dict()
        current.update({k: v for k, v in pretrained.items() if k in current})
        self.model.load_state_dict(current)
        self.model.to(self.device)
        self.model.invalidate_packed_weights()
