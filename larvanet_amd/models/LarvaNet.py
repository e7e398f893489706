"""LarvaNet x4 for MI355X: drop-in for the reference plugin models/LarvaNet.py.

Same plugin surface (create_model(), parse_args/prepare/train_step_larva/upscale/save/restore/...),
same sub-module names and state_dict keys (head.feature_extraction.*, body_{i}.res_blocks.{j}.body.{0,2}.*,
body_{i}.leg.recon_block.{0,2}.*), same initialisation draw order -- but every convolution,
the pixel-shuffle heads, the bicubic base image, the L1 loss and their backward passes run in
the hand-written gfx950 kernels of liblarva_hip.so.  The nn.Conv2d objects below are parameter
containers only; their own forward is never called.

Reference call sites are cited as models/LarvaNet.py:<line> (reference tree).
"""
import argparse
import copy
import os
import time

import numpy as np
import torch
import torch.nn as nn

from .. import dist as ldist
from .. import kernels as K
from ..autograd import (BodyFn, DualChain, ExitFn, ExitsFn, GradBucket, HeadFn, L1LossFn, LegFn, LossTerm, PackedConv, PaddedWidth, mean_of_terms,
                        SideStreams, StepScope, is_large_inference, pack_all)
from ..autograd import step_prologue as autograd_step_prologue
from ..optim import FlatAdamW, flatten_parameters
from ..metrics import image_psnr, image_to_uint8, fit_truth_image_size
from .base import BaseModel

# F.interpolate modes of the base image (models/LarvaNet.py:283-285).  The reference hands the string to
# F.interpolate(x, scale_factor=4, mode=..., align_corners=False) at the first forward; of the modes torch knows
# only these two survive that call (nearest / nearest-exact / area raise ValueError because of align_corners,
# linear / trilinear NotImplementedError for a 4-D input).  Both have a HIP kernel; anything else is refused
# with the same ValueError when the flags are parsed instead of at the first forward.
SUPPORTED_INTERPOLATE = ("bicubic", "bilinear")

NUM_FILTERS = 48  # = 3 * 4**2: PixelShuffle(4) of the leg output must give RGB (models/LarvaNet.py:226,261)
# --num_filters (build-side extension, SURVEY 8a N1: BASELINE configs 2 / 5 name 32- and 64-channel bodies, which the
# reference cannot express): the width of the head's output, the bodies and the legs' first conv; every leg's LAST
# conv keeps 48 outputs, so the exits are still RGB.  48 is the reference's network, bit for bit; other widths have no
# reference counterpart and are checked against oracle/larva_torch.py only.
SUPPORTED_NUM_FILTERS = (32, 48, 64)


def create_model():
    return LarvaNet()


def init_conv(conv, scale=0.1):
    """Reference init (models/LarvaNet.py:22-31): kaiming normal (fan_in, a=0) * scale, zero bias."""
    nn.init.kaiming_normal_(conv.weight, a=0, mode="fan_in")
    conv.weight.data *= scale
    if conv.bias is not None:
        conv.bias.data.zero_()


def _conv(cin, cout):
    return nn.Conv2d(in_channels=cin, out_channels=cout, kernel_size=3, stride=1, padding=1)


def _require_hip(t):
    if not t.is_cuda:
        raise RuntimeError("larvanet_amd: the network only runs on a HIP device (MI355X); "
                           "got a %s tensor and there is no CPU fallback" % t.device)


class _NoScope:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class L1Loss(nn.Module):
    """nn.L1Loss() replacement (models/LarvaNet.py:85) backed by the fused HIP reduction."""

    def forward(self, output, target):
        _require_hip(output)
        return L1LossFn.apply(output, target)


class ResidualBlock(nn.Module):
    """models/LarvaNet.py:205-220"""

    def __init__(self, num_channels):
        super().__init__()
        self.body = nn.Sequential(_conv(num_channels, num_channels), nn.ReLU(inplace=True),
                                  _conv(num_channels, num_channels))
        init_conv(self.body[0])
        init_conv(self.body[2])

    def forward(self, x):
        raise RuntimeError("ResidualBlock is executed by its LarvaBody (fused launch sequence)")


class LarvaHead(nn.Module):
    """models/LarvaNet.py:223-233"""

    def __init__(self, num_filters=NUM_FILTERS):
        super().__init__()
        self.feature_extraction = _conv(3, num_filters)
        init_conv(self.feature_extraction)
        self._pc = PackedConv(self.feature_extraction.weight, self.feature_extraction.bias, cin_pad=16)

    def forward(self, x, x16=None):
        _require_hip(x)
        c = self.feature_extraction
        self._pc.refresh()
        return HeadFn.apply(x.contiguous(), c.weight, c.bias, self._pc, x16, torch.is_grad_enabled())


class LarvaLeg(nn.Module):
    """models/LarvaNet.py:251-267"""

    def __init__(self, num_filters=NUM_FILTERS):
        super().__init__()
        self.recon_block = nn.Sequential(_conv(num_filters, num_filters), nn.ReLU(inplace=True),
                                         _conv(num_filters, NUM_FILTERS))   # 48 = 3 * 4**2 outputs whatever the width
        init_conv(self.recon_block[0])
        init_conv(self.recon_block[2])
        self.upsample = nn.PixelShuffle(4)  # kept for introspection; fused into the conv store
        self._pcs = [PackedConv(self.recon_block[0].weight, self.recon_block[0].bias),
                     PackedConv(self.recon_block[2].weight, self.recon_block[2].bias)]

    def forward(self, fea, base):
        _require_hip(fea)
        c1, c2 = self.recon_block[0], self.recon_block[2]
        for pc in self._pcs:
            pc.refresh()
        return LegFn.apply(fea.contiguous(), base.contiguous(), self._pcs, c1.weight, c1.bias, c2.weight, c2.bias)


class LarvaBody(nn.Module):
    """models/LarvaNet.py:236-248"""

    def __init__(self, num_blocks, num_filters=NUM_FILTERS):
        super().__init__()
        self.res_blocks = nn.Sequential(*[ResidualBlock(num_filters) for _ in range(num_blocks)])
        self.leg = LarvaLeg(num_filters)
        self._pcs = []
        for blk in self.res_blocks:
            self._pcs += [PackedConv(blk.body[0].weight, blk.body[0].bias),
                          PackedConv(blk.body[2].weight, blk.body[2].bias)]

    def forward(self, x):
        _require_hip(x)
        params = []
        for blk in self.res_blocks:
            params += [blk.body[0].weight, blk.body[0].bias, blk.body[2].weight, blk.body[2].bias]
        if not params:
            return x + x
        for pc in self._pcs:
            pc.refresh()
        return BodyFn.apply(x.contiguous(), self._pcs, *params)


def parse_num_blocks(args):
    blocks = [int(v) for v in str(args.num_blocks).split(",")]
    if len(blocks) != args.num_modules:
        # the reference raises GeneratorExit here (models/LarvaNet.py:277-278)
        raise GeneratorExit("Argument num_blocks should have the same number of elements as num_modules.")
    return blocks


class LarvaNetModule(nn.Module):
    """models/LarvaNet.py:270-293"""

    def __init__(self, args):
        super().__init__()
        self.len = args.num_modules
        self.interpolate = args.interpolate
        # inference on widths that are not a multiple of 4: row-padded activations (16-byte LDS-DMA
        # staging) instead of the register-staged conv path; False only to test the latter
        self.pad_odd_widths = True
        self.num_filters = int(getattr(args, "num_filters", NUM_FILTERS))
        if self.num_filters not in SUPPORTED_NUM_FILTERS:
            raise ValueError("larvanet_amd: --num_filters must be one of %s" % (SUPPORTED_NUM_FILTERS,))
        self.head = LarvaHead(self.num_filters)
        for i, nb in enumerate(parse_num_blocks(args)):
            setattr(self, "body_%d" % i, LarvaBody(num_blocks=nb, num_filters=self.num_filters))
        self._join_input_grads()

    def _join_input_grads(self):
        """body i+1's first conv and exit i's first conv read the same tensor: their dgrad weight
        images share an arena so that one launch computes the summed input gradient (JointBwd)."""
        from ..autograd import JointBwd
        for i in range(self.len - 1):
            nxt = getattr(self, "body_%d" % (i + 1))
            if nxt._pcs:
                JointBwd(nxt._pcs[0], getattr(self, "body_%d" % i).leg._pcs[0])

    def packed_convs(self):
        out = []
        for m in self.modules():
            out += ([m._pc] if hasattr(m, "_pc") else []) + list(getattr(m, "_pcs", []))
        return out

    def invalidate_packed_weights(self):
        """Forget the kernel-layout weight images (call after changing weights behind torch's back)."""
        for pc in self.packed_convs():
            pc.invalidate()

    def refresh_packed_weights(self):
        """Rebuild every conv's kernel-layout image with one batched launch (training forward)."""
        pack_all(self.packed_convs())

    def step_prologue(self, x):
        """refresh_packed_weights() + base(x) + the head's padded input as ONE launch -> (base, x16);
        x16 is None when the fused launch does not apply (the head then pads its input itself)."""
        _require_hip(x)
        res = autograd_step_prologue(self.packed_convs(), x) if self.interpolate == "bicubic" else None
        if res is None:
            self.refresh_packed_weights()
            return self.base(x), None
        return res[1], res[0]

    def base(self, x):
        """F.interpolate(x, scale_factor=4, mode=args.interpolate, align_corners=False) (models/LarvaNet.py:283-285)."""
        _require_hip(x)
        if self.interpolate not in SUPPORTED_INTERPOLATE:   # (parse_args already refuses it)
            raise ValueError("larvanet_amd: --interpolate=%s has no HIP kernel; supported: %s"
                             % (self.interpolate, ", ".join(SUPPORTED_INTERPOLATE)))
        with torch.no_grad():
            return K.upsample4(x.detach().contiguous(), self.interpolate)

    def width_scope(self, x):
        """Row-padded activations for inference on widths that are not a multiple of 4."""
        w = int(x.shape[-1])
        if w % 4 and self.pad_odd_widths and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return PaddedWidth(w)
        return _NoScope()

    def forward(self, x):
        with self.width_scope(x):
            base = self.base(x)
            fea = self.head(x)
            for i in range(self.len):
                fea = getattr(self, "body_%d" % i)(fea)
            DualChain.join()   # (no-op unless the layer chain ran as two half-batch chains)
            return getattr(self, "body_%d" % (self.len - 1)).leg(fea, base)


class LarvaNet(BaseModel):
    """Plugin wrapper; control flow of models/LarvaNet.py:42-202."""

    module_class = LarvaNetModule

    def __init__(self):
        super().__init__()
        self.volume_per_step = 0
        self.sync_loss = True
        # sync_loss with a captured step: `return loss.item()` (models/LarvaNet.py:139) waits for the forward only.
        # "poll": one graph; the launch that finishes the loss right after the exits also stores it into a float
        # (+ a sequence number) of coherent pinned host memory (kernels.HostCell), which the host polls.  "split": forward | backward
        # as two graphs, the loss is copied out between them on a side stream and the host waits for that event.
        # False: the host waits for the whole step.
        self.early_loss = {"0": False, "split": "split"}.get(os.environ.get("LARVA_EARLY_LOSS", "poll"), "poll")
        self.use_hip_graph = os.environ.get("LARVA_HIP_GRAPH", "1") != "0"
        self.hip_graph_fell_back = None   # reason, if a capture failed and the step went eager
        # a failed capture raises instead of continuing ~2.4x slower with a printed warning: the drivers
        # (train_larva.py / validate.py / runtime.py) set it unless --allow_eager_fallback, bench.py exits on a fallback
        self.strict_graph = os.environ.get("LARVA_HIP_GRAPH_STRICT", "0") != "0"
        # Exits on a side stream: measured neutral-to-negative on MI355X at batch 16 (same-box A/B:
        # 2.36 ms without, 2.36 / 2.45 ms with, depending on the wgrad variant) -- opt-in.
        self.use_side_streams = os.environ.get("LARVA_SIDE_STREAMS", "0") != "0"
        # weight gradients of all layers in a few large launches at the end of backward
        self.defer_wgrad = os.environ.get("LARVA_DEFER_WGRAD", "1") != "0"
        # one dgrad launch for the two convs that read the same body output (next body + this exit)
        self.joint_input_grads = os.environ.get("LARVA_JOINT_DGRAD", "1") != "0"
        # all exits as one autograd node whose convs go out as batched launches (ExitsFn)
        self.batch_exits = os.environ.get("LARVA_BATCH_EXITS", "1") != "0"
        # the exits' L1 gradient is written by the forward sweep that computes the L1 value
        self.l1_grad_in_forward = os.environ.get("LARVA_L1_GRAD_FWD", "1") != "0"
        # data parallel: all-reduce the first half of the bucket beside the second half's wgrad kernels.  "auto" (the
        # default): decided once at prepare() from a timed isolated all-reduce of the bucket (_choose_dp_schedule)
        self.overlap_allreduce = {"0": False, "1": True}.get(os.environ.get("LARVA_OVERLAP_ALLREDUCE", "auto"), "auto")
        self.dp_schedule = None   # what _choose_dp_schedule measured and chose (bench.py prints it)
        # the body chain as two half-batch chains of strip-tile launches on two streams (autograd.DualChain)
        self.dual_chain = os.environ.get("LARVA_DUAL_CHAIN", "1") != "0"
        # measurement: run the data-parallel step's weight-gradient schedule (two launch groups, so that the first
        # group's slice of the bucket can be all-reduced beside the second) on ONE GPU, without collectives
        self.force_split_backward = os.environ.get("LARVA_FORCE_SPLIT", "0") != "0"

    # ------------------------------------------------------------------ flags
    def _add_args(self, parser):
        parser.add_argument("--num_modules", type=int, default=2, help="Number of cascaded bodies (exits).")
        # the reference declares type=str with an int default, which cannot be split (models/LarvaNet.py:51,276)
        parser.add_argument("--num_blocks", type=str, default="16,16", help="Residual blocks per body, comma separated.")
        parser.add_argument("--interpolate", type=str, default="bicubic", help="Interpolation of the base image.")
        parser.add_argument("--val_volume", type=float, default=30e9, help="Input volume between validations.")
        parser.add_argument("--lr", type=float, default=4e-4, help="Initial learning rate.")
        parser.add_argument("--lr_decay", type=float, default=0.5, help="Learning rate decay factor.")
        parser.add_argument("--lr_step", type=int, default=20000, help="Learning rate decay step (unused, kept for CLI parity).")
        parser.add_argument("--threshold", type=float, default=0.001, help="Plateau threshold (absolute, dB).")
        parser.add_argument("--min_lr", type=float, default=1e-8, help="Minimum learning rate.")
        parser.add_argument("--patience", type=int, default=3, help="Plateau patience.")
        parser.add_argument("--cooldown", type=int, default=6, help="Plateau cooldown.")
        self._add_build_args(parser)

    def _add_build_args(self, parser):
        """Flags the reference does not have (build-side extensions; the defaults are the reference's network)."""
        parser.add_argument("--num_filters", type=int, default=NUM_FILTERS, choices=SUPPORTED_NUM_FILTERS,
                            help="Channels of the head / bodies / legs (the legs' last conv keeps 48). "
                                 "The reference hard-wires 48.")

    def parse_args(self, args):
        parser = argparse.ArgumentParser()
        self._add_args(parser)
        self.args, remaining_args = parser.parse_known_args(args=args)
        if self.args.interpolate not in SUPPORTED_INTERPOLATE:
            raise ValueError("larvanet_amd: --interpolate=%s has no HIP kernel; supported: %s"
                             % (self.args.interpolate, ", ".join(SUPPORTED_INTERPOLATE)))
        return copy.deepcopy(self.args), remaining_args

    # ------------------------------------------------------------------ build
    def _make_scheduler(self):
        return torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optim, mode="max", factor=self.args.lr_decay, patience=self.args.patience,
            cooldown=self.args.cooldown, threshold=self.args.threshold, threshold_mode="abs",
            min_lr=self.args.min_lr)

    def prepare(self, is_training, scales, global_step=0):
        self.global_step = global_step
        self.total_volume = 0.0
        self.temp_volume = 0
        self.scale_list = scales
        for scale in self.scale_list:
            if scale not in (2, 3, 4):
                raise ValueError("Unsupported scale is provided.")
        if len(self.scale_list) != 1:
            raise ValueError("Only one scale should be provided.")
        self.scale = self.scale_list[0]

        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device("cpu")
        self.model = self.module_class(args=self.args).to(self.device)
        ldist.broadcast_parameters(self.model)  # no-op unless torch.distributed is initialised

        if is_training:
            self.loss_fn = L1Loss()
            params = [p for p in self.model.parameters() if p.requires_grad]
            if self.device.type == "cuda":
                # flat buffers: the wgrad kernels write every gradient into one bucket (one
                # all-reduce covers it) and AdamW is one launch over parameters / moments
                flat = flatten_parameters(self.model)
                self.model.invalidate_packed_weights()
                self.grad_bucket = GradBucket(self.model, self.model.packed_convs())
                self.optim = FlatAdamW(params, flat, self.grad_bucket, lr=self.args.lr)
                self._choose_dp_schedule()
            else:
                self.grad_bucket = None
                self.optim = torch.optim.AdamW(params, lr=self.args.lr)
                if self.overlap_allreduce == "auto":   # (no flat bucket on the CPU: nothing to overlap)
                    self.overlap_allreduce = False
                self.dp_schedule = {"choice": "flat", "why": "CPU tensors: one collective over flattened gradients"}
            self.scheduler = self._make_scheduler()

    # Splitting the weight gradients into two launch groups so that the first group's slice of the bucket (its upper
    # ~80 %) can be all-reduced beside the second group's kernels costs a rank a fixed ~36 us per step (two grids + two
    # reductions instead of one flat grid: bench.py's dp_schedule_1gpu, 1.683 against 1.647 ms) and hides at most the
    # first slice's collective under the second group's ~140 us.  With ONE collective after backward the step waits
    # T for it; with the split it waits about 36 + 0.2 T + max(0, 0.8 T - 140).  The split pays for 0.8 T > 36, i.e.
    # T > 45 us if the overlap were perfect; it is not (RCCL's kernels take CUs from the weight-gradient grid), so the
    # rule keeps a factor of two: split when the isolated all-reduce of the bucket takes more than 90 us.
    DP_SPLIT_ABOVE_US = 90.0

    def _choose_dp_schedule(self):
        """LARVA_OVERLAP_ALLREDUCE=auto: time the bucket's all-reduce once and pick the schedule.
        `dp_schedule["choice"]` names the schedule the step really runs (LARVA_FORCE_SPLIT included), and
        `overlap_allreduce` is a bool on every path out of here."""
        ws = ldist.world_size()
        if self.overlap_allreduce != "auto":
            self.overlap_allreduce = bool(self.overlap_allreduce)
            self.dp_schedule = {"why": "LARVA_OVERLAP_ALLREDUCE"}
        elif not ldist.active():
            self.overlap_allreduce = True     # (only consulted with more than one rank, or by LARVA_FORCE_SPLIT)
            self.dp_schedule = {"why": "one rank, no communicator"}
        else:
            t_us = ldist.time_allreduce_us(self.grad_bucket.flat)
            self.overlap_allreduce = t_us > self.DP_SPLIT_ABOVE_US
            self.dp_schedule = {"allreduce_isolated_us": t_us, "bucket_bytes": int(self.grad_bucket.flat.numel()) * 4, "ranks": ws,
                                "rule": "split (two weight-gradient launch groups, the first slice all-reduced beside the second) "
                                        "when the isolated all-reduce of the bucket takes more than %.0f us" % self.DP_SPLIT_ABOVE_US}
        self.dp_schedule["choice"] = "split" if self._split_backward() else "flat"
        if self.force_split_backward:
            self.dp_schedule["forced"] = "LARVA_FORCE_SPLIT"
        if ldist.active() and ldist.is_main() and "allreduce_isolated_us" in self.dp_schedule:
            print("data-parallel weight-gradient schedule: %s (isolated all-reduce of the %.2f MB bucket over %d ranks: %.1f us)"
                  % (self.dp_schedule["choice"], self.dp_schedule["bucket_bytes"] / 1e6, ws, self.dp_schedule["allreduce_isolated_us"]))

    # ------------------------------------------------------------------ training
    def _grad_one(self, loss):
        """d loss / d loss = 1 as a persistent tensor (autograd would fill a new one every step)."""
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device:
            one = self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        return one

    def _num_loss_terms(self):
        return self.args.num_modules

    def _exit_fused(self, leg, fea, base, truth_tensor):
        """leg(fea, base) and loss_fn(out, truth) as one autograd node (ExitFn) when loss_fn is
        the stock L1Loss; otherwise the two separate calls of the reference.  Returns (image, term):
        in the fused case the term is a LossTerm of partial sums that the mean over the exits
        finishes (no per-exit finishing launch, the 1/M of the mean applied inside L1's backward)."""
        if isinstance(self.loss_fn, L1Loss) and isinstance(leg, LarvaLeg):
            for pc in leg._pcs:
                pc.refresh()
            c1, c2 = leg.recon_block[0], leg.recon_block[2]
            out, part = ExitFn.apply(fea.contiguous(), base.contiguous(), truth_tensor.contiguous(), leg._pcs,
                                     c1.weight, c1.bias, c2.weight, c2.bias, self._num_loss_terms())
            return out, LossTerm(part, 1.0 / float(out.numel()), prescaled=True)
        out = leg(fea, base)
        return out, self.loss_fn(out, truth_tensor)

    def _exit(self, leg, fea, base, truth_tensor):
        """One exit (leg + its L1 term); on the `leg` side stream when side streams are active, so
        that it overlaps the next body (forward) and the previous body's backward."""
        if not SideStreams.active:
            return self._exit_fused(leg, fea, base, truth_tensor)
        side = SideStreams.fork("leg", fea, base, truth_tensor)
        with torch.cuda.stream(side):
            out, term = self._exit_fused(leg, fea, base, truth_tensor)
        SideStreams.keep(out, term.tensor if isinstance(term, LossTerm) else term)
        self._pending_exit_sync = True
        return out, term

    def _sync_exits(self):
        if getattr(self, "_pending_exit_sync", False):
            torch.cuda.current_stream().wait_stream(SideStreams.get("leg"))
            self._pending_exit_sync = False

    def _exits_batched(self):
        """All exits as one autograd node with batched launches (ExitsFn): the stock L1 loss on
        stock legs, training-shaped input (no row pitch), no side streams."""
        return (self.batch_exits and isinstance(self.loss_fn, L1Loss) and not SideStreams.active
                and PaddedWidth.current is None
                and all(isinstance(getattr(self.model, "body_%d" % i).leg, LarvaLeg) for i in range(self.args.num_modules)))

    def _all_exits(self, feas, base, truth_tensor):
        """(last exit's image, [LossTerm per exit]) from the body outputs, in one ExitsFn node."""
        legs, params = [], []
        for i in range(len(feas)):
            leg = getattr(self.model, "body_%d" % i).leg
            for pc in leg._pcs:
                pc.refresh()
            legs.append(leg._pcs)
            c1, c2 = leg.recon_block[0], leg.recon_block[2]
            params += [c1.weight, c1.bias, c2.weight, c2.bias]
        res = ExitsFn.apply(base.contiguous(), truth_tensor.contiguous(), legs, self._num_loss_terms(),
                            *[f.contiguous() for f in feas], *params)
        scale = 1.0 / float(res[0].numel())
        return res[0], [LossTerm(p, scale, prescaled=True) for p in res[1:]]

    def _exit_losses(self, input_tensor, truth_tensor):
        """Forward through every exit (models/LarvaNet.py:102-109). Returns (loss, last output)."""
        net = self.model
        # weight images + bicubic base + the head's padded input in one launch, before the layer chain forks
        base, x16 = net.step_prologue(input_tensor)
        fea = net.head(input_tensor, x16)
        if self._exits_batched():
            # the exits do not feed the bodies: run the body chain first, then all exits together
            feas = []
            for i in range(self.args.num_modules):
                fea = getattr(net, "body_%d" % i)(fea)
                feas.append(fea)
            DualChain.join()   # the two half-batch chains meet again: the exits read whole tensors
            out, terms = self._all_exits(feas, base, truth_tensor)
            return mean_of_terms(terms), out
        terms = []
        out = None
        for i in range(self.args.num_modules):
            body = getattr(net, "body_%d" % i)
            fea = body(fea)
            out, term = self._exit(body.leg, fea, base, truth_tensor)
            terms.append(term)
        self._sync_exits()
        return mean_of_terms(terms), out

    # hipGraph path: one step issues ~330 short kernels; launched one by one from Python the GPU
    # idles between them, so forward + backward are captured once per batch shape and replayed.
    def _graph_key(self, input_tensor, truth_tensor):
        return (tuple(input_tensor.shape), tuple(truth_tensor.shape), str(input_tensor.device), self._early_loss_capture())

    def _early_loss_capture(self):
        if not (self.sync_loss and self.early_loss):
            return False
        return "split" if self.early_loss == "split" else "poll"

    def _scope(self, early_loss=False):
        # seed_grad: _forward_backward seeds loss.backward() with _grad_one and nothing scales the loss
        # chains are joined only after the body loop (forward) / at the end of backward when nothing
        # else reads a chain tensor in between: batched exits after the bodies; and in backward one
        # consumer per body output (joint input gradients) with every weight gradient deferred
        lazy_fwd = self._exits_batched()
        lazy_bwd = (lazy_fwd and self.joint_input_grads and self.defer_wgrad and self._single_consumer_features()
                    and getattr(self, "grad_bucket", None) is not None and self.grad_bucket.intact(self.model))
        return StepScope(side_streams=self.use_side_streams, defer_wgrad=self.defer_wgrad,
                         split_flush=self._split_backward(), joint_input_grads=self.joint_input_grads,
                         seed_grad=1.0 if self.l1_grad_in_forward else None,
                         dual_chain=self.dual_chain and not self.use_side_streams, lazy_chain_joins=(lazy_fwd, lazy_bwd),
                         early_loss=early_loss)

    def _single_consumer_features(self):
        """Is every body output read by its exit and the next body only (V2's tail reads them too)?"""
        return True

    def _split_backward(self):
        """Data parallel with in-place gradients: backward ends in two halves so that the
        all-reduce of the first overlaps the weight-gradient kernels of the second."""
        bucket = getattr(self, "grad_bucket", None)
        return (self.overlap_allreduce is True and self.defer_wgrad and (ldist.active() or self.force_split_backward)
                and bucket is not None and bucket.intact(self.model))

    def _note_early(self, scope):
        """Where the gradients that are complete after the first half live in the flat bucket:
        self._early_lo = first float of that suffix, or None = no overlap (one collective)."""
        self._early_lo = None
        from ..autograd import DeferredWgrad
        bucket = getattr(self, "grad_bucket", None)
        if not scope.split_flush or bucket is None or not scope.early_targets:
            return
        early, late = bucket.span(scope.early_targets), bucket.span(DeferredWgrad.late_targets())
        if early is None or late is None:
            return
        if early[1] == bucket.flat.numel() and late[1] <= early[0]:
            self._early_lo = early[0]

    def _capture_step(self, input_tensor, truth_tensor):
        from ..autograd import DeferredWgrad
        self._static_in = input_tensor.clone()
        self._static_truth = truth_tensor.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up outside capture (lazy kernel attributes, allocator pools)
                self._zero_grad()
                with self._scope():
                    loss, _ = self._exit_losses(self._static_in, self._static_truth)
                    loss.backward(self._grad_one(loss))
                DeferredWgrad.flush_late()
        torch.cuda.current_stream().wait_stream(side)
        self._zero_grad()
        graph = torch.cuda.CUDAGraph()
        self._graph_back = None
        # thread_local: a process-group watchdog thread must not abort the capture
        mode = self._early_loss_capture()
        if mode == "split" and getattr(self, "_loss_host", None) is None:
            self._loss_host = torch.empty((), dtype=torch.float32).pin_memory()
            self._loss_stream = torch.cuda.Stream()
            self._loss_done = torch.cuda.Event()
            self._fwd_done = torch.cuda.Event()
        if mode == "poll" and getattr(self, "_loss_cell", None) is None:
            self._loss_cell = K.HostCell()
        if mode != "split":
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                with self._scope(early_loss=self._loss_cell if mode else False) as scope:
                    loss, out = self._exit_losses(self._static_in, self._static_truth)
                    loss.backward(self._grad_one(loss))
        else:
            # forward | backward as two graphs over one memory pool: the loss is complete when the first one ends
            scope = self._scope(early_loss=True)
            back = torch.cuda.CUDAGraph()
            scope.__enter__()
            left = False
            try:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss, out = self._exit_losses(self._static_in, self._static_truth)
                    DualChain.join()
                with torch.cuda.graph(back, pool=graph.pool(), capture_error_mode="thread_local"):
                    loss.backward(self._grad_one(loss))
                    left = True
                    scope.__exit__(None, None, None)   # joins the chains, issues the queued weight gradients
            finally:
                if not left:
                    scope.__exit__(RuntimeError, None, None)
            self._graph_back = back
        self._graph_polls = mode == "poll"
        self._note_early(scope)
        self._graph_late = None
        if DeferredWgrad._late:  # second half of a split backward: its own graph, same memory pool
            late = torch.cuda.CUDAGraph()
            with torch.cuda.graph(late, pool=graph.pool(), capture_error_mode="thread_local"):
                DeferredWgrad.flush_late()
            self._graph_late = late
        self._graph, self._graph_loss, self._graph_out = graph, loss, out
        self._graph_shape = self._graph_key(input_tensor, truth_tensor)

    def input_buffers(self, input_shape, truth_shape):
        """The (input, truth) tensors the captured training step reads, or None while no step of
        that shape has been captured.  A producer on the device (dataloaders/device_patch_loader)
        writes the next batch straight into them and passes them to train_step_larva, which then
        skips its two copies into the graph's inputs."""
        if not self.use_hip_graph or getattr(self, "_graph_shape", None) is None:
            return None
        if tuple(self._static_in.shape) != tuple(input_shape) or tuple(self._static_truth.shape) != tuple(truth_shape):
            return None
        return self._static_in, self._static_truth

    def _stage_inputs(self, input_tensor, truth_tensor):
        """The batch into the captured step's input buffers (train_larva.py:123-128 hands over fresh device tensors every
        step), on the current stream, i.e. ordered behind whatever produced them.  (Round 4 also copied them on a stream
        of its own beside the previous step's backward: 1.669-1.672 against 1.657-1.660 ms -- two cross-stream waits cost
        more than the two 5 us copies they hide -- and, without an edge from the producer's stream, a race; removed.)"""
        for dst, src in ((self._static_in, input_tensor), (self._static_truth, truth_tensor)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)

    def _zero_grad(self):
        """optim.zero_grad() of the reference (models/LarvaNet.py:112).  With the flat gradient
        bucket every backward overwrites the gradients in place, so nothing has to be cleared
        (and set_to_none would detach the bucket views)."""
        bucket = getattr(self, "grad_bucket", None)
        if bucket is not None and bucket.intact(self.model):
            return
        if bucket is not None:  # somebody replaced a .grad: stop writing in place
            for pc in self.model.packed_convs():
                pc.grad_inplace = False
            self.grad_bucket = None
        self.optim.zero_grad(set_to_none=True)

    def _forward_backward(self, input_tensor, truth_tensor):
        """loss and gradients of one batch (models/LarvaNet.py:101-113)."""
        if self.use_hip_graph and input_tensor.is_cuda:
            if getattr(self, "_graph_shape", None) != self._graph_key(input_tensor, truth_tensor):
                try:
                    self._capture_step(input_tensor, truth_tensor)
                except Exception as e:  # capture is an optimisation: fall back to plain launches
                    if self.strict_graph:
                        raise
                    print("WARNING: hipGraph capture failed (%s: %s); continuing with eager launches"
                          % (type(e).__name__, e))
                    self.use_hip_graph = False
                    self.hip_graph_fell_back = "%s: %s" % (type(e).__name__, e)
                    torch.cuda.synchronize()
                    DualChain.reset()   # (a capture that died mid-chain must not leave the chains marked as forked)
                    return self._forward_backward(input_tensor, truth_tensor)
            # (a producer that filled input_buffers() in place hands the very same storage back)
            self._stage_inputs(input_tensor, truth_tensor)
            if self._graph_polls:
                self._loss_cell.expect()   # this replay's store carries the next sequence number
                self._loss_in_flight = "poll"
            self._graph.replay()  # gradients are overwritten in place: no zero_grad needed
            if self._graph_back is not None:
                # the loss goes to pinned host memory on a stream of its own while backward runs
                self._fwd_done.record()
                with torch.cuda.stream(self._loss_stream):
                    self._loss_stream.wait_event(self._fwd_done)
                    self._loss_host.copy_(self._graph_loss, non_blocking=True)
                    self._loss_done.record()
                self._loss_in_flight = "event"
                self._graph_back.replay()
            self._late = self._graph_late.replay if self._graph_late is not None else None
            return self._graph_loss, self._graph_out
        from ..autograd import DeferredWgrad
        self._zero_grad()
        with self._scope() as scope:
            loss, out = self._exit_losses(input_tensor, truth_tensor)
            loss.backward(self._grad_one(loss))
        self._note_early(scope)
        self._late = DeferredWgrad.flush_late if DeferredWgrad._late else None
        return loss, out

    def _finish_backward(self):
        """Second half of a split backward + the data-parallel mean of the gradients
        (one flat bucket; the all-reduce of the half that is already complete runs on
        RCCL's stream beside the remaining weight-gradient kernels).  The 1/world_size of the
        mean is applied inside the optimizer kernel (FlatAdamW.mean_scale)."""
        late, self._late = getattr(self, "_late", None), None
        ws = ldist.world_size()
        bucket = getattr(self, "grad_bucket", None)
        if not ldist.active() or bucket is None or not bucket.intact(self.model):
            if late is not None:
                late()
            ldist.allreduce_gradients(self.model, None)   # already the mean: no second 1/world in the optimizer
            if isinstance(self.optim, FlatAdamW):
                self.optim.mean_scale = 1.0
            return
        lo = getattr(self, "_early_lo", None)
        timed = getattr(self, "time_allreduce", False) and self.device.type == "cuda"
        if late is None or lo is None or lo <= 0:
            if late is not None:
                late()
            t0 = self._mark(timed)
            ldist.allreduce_sum(bucket.flat)
        else:
            work = ldist.allreduce_sum(bucket.flat[lo:], async_op=True)
            late()
            t0 = self._mark(timed)   # the last weight-gradient kernel has been issued: what follows is exposed
            ldist.allreduce_sum(bucket.flat[:lo])
            work.wait()
        if timed:
            self.allreduce_events.append((t0, self._mark(True)))
        if isinstance(self.optim, FlatAdamW):
            self.optim.mean_scale = 1.0 / ws
        else:
            bucket.flat.mul_(1.0 / ws)

    def _mark(self, on):
        """Timing event on the current stream (bench.py: exposed all-reduce time per step = from the
        end of the last weight-gradient kernel to the point where AdamW may start)."""
        if not on:
            return None
        if not hasattr(self, "allreduce_events"):
            self.allreduce_events = []
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def train_step_larva(self, args, val_dataloader, input_tensor, truth_tensor, summary=None):
        self.global_step += 1
        self.temp_volume += self.volume_per_step

        loss, out = self._forward_backward(input_tensor, truth_tensor)
        self._finish_backward()  # rest of a split backward + mean of the gradients over ranks
        loss_copy = None
        if not self.sync_loss and isinstance(self.optim, FlatAdamW) and loss.is_cuda:
            # the caller gets a tensor of its own (the captured step's loss buffer is overwritten by the next
            # replay); the optimizer launch makes the 4-byte copy
            loss_copy = torch.empty_like(loss)
            self.optim.copy_scalar = (loss.detach(), loss_copy)
        self.optim.step()
        if loss_copy is not None and self.optim.copy_scalar is not None:   # the step took another path: copy here
            self.optim.copy_scalar = None
            loss_copy = None
        self.model.invalidate_packed_weights()  # the kernel-layout weight images are now stale

        if self.global_step == 1:
            self.validate_for_train(args, val_dataloader)

        if self.temp_volume >= self.args.val_volume:
            self.total_volume += self.temp_volume
            self.temp_volume = 0
            self.validate_for_train(args, val_dataloader)
            if ldist.is_main():
                self.save(base_path=args.train_path)
                if getattr(args, "save_train_state", False):
                    self.save_training_state(base_path=args.train_path)
                print(f"saved a model checkpoint at volume {self.total_volume/1e9:.0f}G")
            if summary is not None:
                self._write_summary(summary, loss, input_tensor, out, truth_tensor)

        # the reference returns loss.item() (a host sync every step, models/LarvaNet.py:139);
        # sync_loss=False hands back a 0-d device tensor instead so the host can run ahead (a copy:
        # the captured step's own loss tensor is overwritten by the next replay)
        if self.sync_loss:
            how, self._loss_in_flight = getattr(self, "_loss_in_flight", False), False
            if how == "event":   # (early-loss captures: see _forward_backward)
                self._loss_done.synchronize()
                return self._loss_host.item()
            if how == "poll":
                return self._poll_loss()
            return loss.item()
        self._loss_in_flight = False
        return loss_copy if loss_copy is not None else loss.detach().clone()

    def _poll_loss(self):
        """The step's loss as soon as the launch that finishes it has stored it into the host cell.  The store carries a
        sequence number, so a late store of an earlier replay is never taken for this one's (and a loss that IS NaN is
        just a value).  A short spin -- the loss is normally there within the forward's ~0.6 ms --, then the core is
        yielded between looks; after 5 s the stream is synchronised (the launch must then have stored)."""
        cell = self._loss_cell
        t0 = time.perf_counter()
        spins = 0
        while True:
            v = cell.take()
            if v is not None:
                return v
            spins += 1
            if spins > 2000:   # ~1 ms of looking without a break
                dt = time.perf_counter() - t0
                if dt > 5.0:
                    break
                time.sleep(0 if dt < 0.02 else 0.0005)
        torch.cuda.current_stream().synchronize()
        v = cell.take()
        if v is None:
            raise RuntimeError("larvanet_amd: the captured step finished without storing its loss "
                               "(host cell at sequence %d, expected %d)" % (cell.sequence, cell.expected))
        return v

    def _write_summary(self, summary, loss, input_tensor, out, truth_tensor):
        summary.add_scalar("loss", loss, self.global_step)
        summary.add_scalar("lr", self.get_lr(), self.global_step)
        tensors = {"input": input_tensor, "output": out.detach(), "truth": truth_tensor}
        for name, t in tensors.items():
            t8 = t.clamp(0, 255).byte()
            for i in range(min(4, len(t8))):
                summary.add_image("%s/%d" % (name, i), t8[i], self.global_step)

    def validate_for_train(self, args, dataloader):
        """Mean RGB PSNR over the validation set drives ReduceLROnPlateau (models/LarvaNet.py:141-161).
        Under data parallelism rank r scores images r, r+world, ... and the sum is all-reduced so
        that every rank steps its scheduler with the same value."""
        print("begin validation")
        num_images = dataloader.get_num_images()
        psnr_sum = 0.0
        with torch.no_grad():
            for image_index in range(ldist.rank(), num_images, ldist.world_size()):
                input_image, truth_image, _ = dataloader.get_image_pair(image_index=image_index, scale=4)
                psnr_sum += self.image_psnr(input_image, truth_image)
        average_psnr = ldist.allreduce_scalar_sum(psnr_sum, self.device) / max(num_images, 1)
        print(f"step {self.global_step}, volume {self.total_volume/1e9:.0f}G,"
              f" psnr={average_psnr:.8f}, lr = {self.get_lr():.8f}")
        self.scheduler.step(average_psnr)
        return average_psnr

    def image_psnr(self, input_image, truth_image):
        """PSNR of one validation pair under the validate.py protocol (uint8 round/clip, truth
        cropped top-left, all RGB pixels).  On a HIP device the conversion and the squared error
        run in one kernel and only 8 bytes come back; the numbers equal the host protocol's."""
        if self.device.type == "cuda":
            out = self.model(self._to_input_tensor([input_image]))[0].contiguous()
            truth8 = torch.from_numpy(np.ascontiguousarray(image_to_uint8(truth_image))).to(self.device)
            return K.psnr_u8(out, truth8)
        output_image = image_to_uint8(self.upscale(input_list=[input_image], scale=4)[0])
        truth8 = fit_truth_image_size(output_image=output_image, truth_image=image_to_uint8(truth_image))
        return float(image_psnr(output_image=output_image, truth_image=truth8))

    # ------------------------------------------------------------------ inference
    def _infer_scope(self):
        """Inference forward inside a captured graph: the layer chain may run as two half-batch chains."""
        return StepScope(defer_wgrad=False, joint_input_grads=False, dual_chain=self.dual_chain,
                         lazy_chain_joins=(True, False))

    def _infer(self, x):
        """self.model(x) without gradients.  A batch shape seen for the second time is captured into a
        hipGraph (launched one by one from Python the ~36 kernels of a 16 x 3 x 48 x 48 forward are
        host-bound: 0.70 ms against 0.5 ms of GPU time) and replayed from then on; shapes seen once --
        validation images all differ in size -- run eagerly.  The returned tensor of a replay is the
        graph's output buffer: callers that keep it across calls copy it (upscale() moves it to the
        host anyway)."""
        if not (self.use_hip_graph and x.is_cuda) or torch.is_grad_enabled():
            return self.model(x)
        # A whole validation image is 36 launches of 60 us each: the host is ~2 ms ahead of the GPU after the first few,
        # and eager launches have no replay boundary and no copy into a static input: 2.162 against 2.178 ms per
        # 339 x 510 image (tools/infer_modes.py, round 5).  The capture pays where the launches are short.
        if is_large_inference(x.shape[0], x.shape[2], x.shape[3]):   # (the same rule picks the direct head kernel)
            return self.model(x)
        cache = self.__dict__.setdefault("_infer_graphs", {})
        seen = self.__dict__.setdefault("_infer_seen", {})
        key = tuple(x.shape)
        ent = cache.get(key)
        if ent is None:
            if len(seen) > 512:   # (a long run over images of ever new sizes: forget the counts)
                seen.clear()
            seen[key] = seen.get(key, 0) + 1
            if seen[key] < 2 or len(cache) >= 4:
                return self.model(x)
            ent = cache[key] = self._capture_infer(x)
            if ent is False:
                return self.model(x)
        if ent is False:
            return self.model(x)
        for pc in self.model.packed_convs():   # weights restored / stepped since the capture: repack (outside the graph)
            pc.refresh()
        static_x, graph, out = ent
        static_x.copy_(x)
        graph.replay()
        return out

    def _capture_infer(self, x):
        static_x = x.clone()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    with self._infer_scope():
                        self.model(static_x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                with self._infer_scope():
                    out = self.model(static_x)
            return static_x, graph, out
        except Exception as e:   # an optimisation only
            if self.strict_graph:
                raise
            print("WARNING: hipGraph capture of the inference forward failed (%s: %s); running it eagerly"
                  % (type(e).__name__, e))
            torch.cuda.synchronize()
            return False

    def _to_input_tensor(self, input_list):
        arr = np.ascontiguousarray(np.stack([np.asarray(a, dtype=np.float32) for a in input_list]))
        return torch.from_numpy(arr).to(self.device)

    def upscale(self, input_list, scale):
        """list of CHW numpy images -> (N, 3, 4H, 4W) float32 numpy (models/LarvaNet.py:163-171)."""
        with torch.no_grad():
            return self._infer(self._to_input_tensor(input_list)).detach().cpu().numpy()

    def upscale_tensor(self, input_list):
        """upscale() without the trip to the host: (N, 3, 4H, 4W) float32 on self.device."""
        with torch.no_grad():
            return self._infer(self._to_input_tensor(input_list)).detach().clone()

    def receptive_halo(self):
        """LR pixels beyond an output pixel's own LR pixel that can influence it: one per 3x3
        convolution on the deepest path (head + 2 per residual block + 2 in the leg; the bicubic
        base needs 2).  Sub-images cut with this halo reproduce the full image exactly
        (image_utils.upscale_band)."""
        return 1 + 2 * sum(parse_num_blocks(self.args)) + 2

    def test(self, input_list):
        if torch.is_grad_enabled():
            return self.model(self._to_input_tensor(input_list))
        return self._infer(self._to_input_tensor(input_list)).clone()

    def fwd_runtime(self, input_tensor):
        """models/LarvaNet.py:200-202; under torch.no_grad() a repeated shape replays a captured graph and
        the result is that graph's output buffer (overwritten by the next call)."""
        if torch.is_grad_enabled():
            return self.model(input_tensor)
        return self._infer(input_tensor)

    # ------------------------------------------------------------------ checkpoints
    def save(self, base_path):
        """Bare state_dict with the reference's key names (models/LarvaNet.py:183-185)."""
        save_path = os.path.join(base_path, "model_step%d_vol%.0fG.pth" % (self.global_step, self.total_volume / 1e9))
        torch.save({k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}, save_path)
        return save_path

    def restore(self, ckpt_path, target=None):
        self.model.load_state_dict(torch.load(ckpt_path, map_location=self.device))
        self.model.invalidate_packed_weights()

    def save_training_state(self, base_path):
        """Optimizer moments, LR-scheduler state, step / volume counters and RNG states next to
        the reference-compatible weight file (the reference can only resume weights + --global_step,
        models/LarvaNet.py:183-188, train_larva.py:43-47,81-83)."""
        path = os.path.join(base_path, "train_state_step%d.pth" % self.global_step)
        opt = self.optim.training_state() if hasattr(self.optim, "training_state") else \
            {"flat": False, "torch": self.optim.state_dict()}
        torch.save({"global_step": self.global_step, "total_volume": self.total_volume,
                    "temp_volume": self.temp_volume, "optimizer": opt, "scheduler": self.scheduler.state_dict(),
                    "torch_rng": torch.get_rng_state(), "numpy_rng": np.random.get_state()}, path)
        return path

    def restore_training_state(self, path):
        st = torch.load(path, map_location="cpu", weights_only=False)
        self.global_step, self.total_volume, self.temp_volume = st["global_step"], st["total_volume"], st["temp_volume"]
        if hasattr(self.optim, "load_training_state"):
            self.optim.load_training_state(st["optimizer"])
        else:
            self.optim.load_state_dict(st["optimizer"]["torch"])
        self.scheduler.load_state_dict(st["scheduler"])
        torch.set_rng_state(st["torch_rng"])
        np.random.set_state(st["numpy_rng"])

    def get_model(self):
        return self.model

    def get_next_train_scale(self):
        return self.scale_list[np.random.randint(len(self.scale_list))]

    def get_lr(self):
        return self.optim.param_groups[0]["lr"]
