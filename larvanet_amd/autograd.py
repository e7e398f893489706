"""torch.autograd glue: the reference trains with `loss.backward()` (models/LarvaNet.py:113), so
the HIP forward/backward passes are exposed as autograd Functions at the granularity of the
reference's sub-modules (head, body_i, body_i.leg, tail, loss).  No arithmetic happens here --
only the order of kernel launches and which activations are kept for the backward pass.

Backward structure of one residual block  y = x + conv2(relu(conv1(x)))  given g = dL/dy:
    dh = dgrad(conv2)(g) * [h > 0]          (mask fused into the conv epilogue)
    dx = g + dgrad(conv1)(dh)               (skip gradient fused as a residual operand)
    dW2, db2 = wgrad(g, h);  dW1, db1 = wgrad(dh, x)
Inside the plugin's training step (StepScope) the weight gradients of ALL layers are queued and
issued in two or three large launches when backward ends (DeferredWgrad), all exits run as one node
with batched launches (ExitsFn), and a tensor read by two convs gets its gradient from one stacked
dgrad launch (JointBwd / JointInputGrad).
"""
import os

import numpy as np
import torch

from . import kernels as K

# Workgroups per weight-gradient launch: one per CU of an MI355X.
_WGRAD_WORKGROUPS = 256


class PaddedWidth:
    """Inference on images whose width W is not a multiple of 4 (DIV2K x4 LR images are 510 wide):
    inside `with PaddedWidth(W):` every LR-resolution activation is allocated with its rows padded
    to P = W rounded up to 4 and columns [W, P) kept at zero, so the convs keep their 16-byte
    LDS-DMA staging path instead of the scalar-load fallback.  Forward only."""

    current = None

    def __init__(self, w):
        self.w = int(w)

    def __enter__(self):
        self._prev = PaddedWidth.current
        PaddedWidth.current = self.w
        return self

    def __exit__(self, *exc):
        PaddedWidth.current = self._prev
        return False

    @staticmethod
    def pitch_of(w):
        return (int(w) + 3) // 4 * 4


def _lw():
    return PaddedWidth.current


class SideStreams:
    """Concurrency inside one training step.  A conv launch keeps the matrix pipes busy only
    about half of its duration (tile staging, the store burst and the launch floor are exposed),
    and the conv / wgrad kernels are sized so that two workgroups share a CU -- so independent
    work is put on side streams and the hardware overlaps it:
      * `leg`:   exit i (leg convs + L1) runs beside body i+1, forward and -- because autograd
                 replays a node on its forward stream -- backward;
      * `wgrad`: weight-gradient batches (never on the critical path) run beside the dgrad chain.
    Only active inside `with StepScope(side_streams=True):` (the plugin's forward+backward), which joins
    every side stream before it returns; anywhere else everything stays on the current stream.
    Tensors that a side stream still reads are kept referenced until the join, so the caching
    allocator cannot hand their memory out early (also during hipGraph capture)."""

    active = False
    wgrad_on_side = False   # measured: wgrad alone already runs at 95 % of the MFMA peak and only
                            # slows the conv chain down when it shares the CUs with it
    _streams = {}
    _keep = []

    @classmethod
    def get(cls, name):
        dev = torch.cuda.current_device()
        key = (dev, name)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=dev)
        return cls._streams[key]

    @classmethod
    def fork(cls, name, *tensors):
        """Side stream `name`, ordered after everything issued so far on the current stream."""
        side = cls.get(name)
        side.wait_stream(torch.cuda.current_stream())
        cls._keep.extend(tensors)
        return side

    @classmethod
    def keep(cls, *tensors):
        cls._keep.extend(tensors)

    @classmethod
    def join(cls):
        cur = torch.cuda.current_stream()
        for (dev, _), st in cls._streams.items():
            if dev == torch.cuda.current_device():
                cur.wait_stream(st)
        cls._keep.clear()


class DualChain:
    """The dependent layer chain of a step (head -> 32 body convs forward, 32 dgrads backward) as TWO
    independent chains, one per half of the batch, on two streams.

    Why: at the training shape a conv launch is 256 workgroups = one per CU, and a lone workgroup
    cannot hide its own prologue (argument fetch, first LDS-DMA chunks), store burst and the launch
    boundary: 15.4 us per layer against 13.2 us of MFMA-paced K loop.  Images are independent, so the
    two halves of the batch are two independent chains; with the conv kernel's strip tiles (5 x 16 /
    4 x 16 pixels instead of 3 x 48, kernels.conv3x3(strips=...)) each half-batch layer is again a
    256-workgroup launch, the two chains' launches share every CU two by two and run out of phase.
    The second chain's tile table starts with the other tile height, so the two workgroups that meet
    on a CU are a 5-row and a 4-row tile (4 + 3 MFMAs per k-step per SIMD, the 3 x 48 tile's 7):
    13.8 us per full-batch layer (tools/bench_dual_chain.py).  Results are bit-identical to the
    single chain (the K loop of every output pixel is unchanged).

    Streams fork from the current stream at the first chained conv and join it again in join(): in
    `lazy` mode (the plugin's step, whose graph has no other consumer in between) only where the
    caller says so, otherwise at the end of every autograd node that used the chains.  Every tensor
    a side stream touches is kept referenced until the join (the caching allocator would otherwise
    hand its memory to a later allocation on the main stream)."""

    enabled = False      # set by StepScope
    # output store policy of the strip launches: "fwd" = plain stores in the forward chain (read at once by
    # the next launch), non-temporal in the backward chain; "all" / "none" for A/B timing
    # round 4, same box, three alternating rounds (profiles/r04_ab_strip_plain.txt): all 1.634-1.641 ms, fwd 1.648-1.650,
    # none 1.657-1.660 -- round 2's A/B (all-plain 1.696 against forward-only 1.688) no longer holds for today's kernels
    plain_stores = os.environ.get("LARVA_STRIP_PLAIN", "all")
    lazy_fwd = False
    lazy_bwd = False
    max_workgroups = 512   # one full-batch launch with more 3 x 48 tiles than this already overlaps by itself
    _streams = {}
    _forked = False
    _keep = []

    @classmethod
    def wants(cls, n, h, p):
        """Should a conv over [n][*][h][p] run as two half-batch strip-tile launches?"""
        if not cls.enabled or n < 2 or p % 4:
            return False
        if n * ((h + 2) // 3) * ((p + 47) // 48) > cls.max_workgroups:
            return False
        return K.strip_tile_table(h, p, torch.device("cuda", torch.cuda.current_device())) is not None

    # Round 4: chain 0 runs on the stream that was current at the fork (the capture's own stream) and only chain 1 on a
    # side stream: one cross-stream edge per fork and per join instead of two (each costs the dependent launch ~3-10 us,
    # tools/step_marks.py).  LARVA_CHAIN0_ON_MAIN=0: both chains on side streams (rounds 2-3).
    chain0_on_main = os.environ.get("LARVA_CHAIN0_ON_MAIN", "1") != "0"
    _main = None

    @classmethod
    def _stream(cls, k):
        if k == 0 and cls.chain0_on_main and cls._main is not None:
            return cls._main
        key = (torch.cuda.current_device(), k)
        if key not in cls._streams:
            cls._streams[key] = torch.cuda.Stream(device=key[0])
        return cls._streams[key]

    @classmethod
    def conv(cls, srcs, wpk, cout, forward=False, **kw):
        """kernels.conv3x3 for a link of the chain (same arguments, `out` allocated here).  forward: a
        link of the forward chain (its output is read by the next launch at once: plain stores)."""
        first = srcs if isinstance(srcs, torch.Tensor) else srcs[0]
        n, _, h, p = (int(v) for v in first.shape)
        if kw.get("logical_w") is not None or kw.get("shuffle") or not cls.wants(n, h, p):
            if cls._forked:      # a link that does not split: back to one stream first
                cls.join()
            return K.conv3x3(srcs, wpk, cout, **kw)
        out = torch.empty((n, cout, h, p), device=first.device, dtype=torch.float32)
        cur = torch.cuda.current_stream()
        if not cls._forked:
            cls._main = cur
            for k in range(2):
                if cls._stream(k) != cur:
                    cls._stream(k).wait_stream(cur)
            cls._forked = True
        cls._keep.append(out)
        cls._keep.extend([srcs] if isinstance(srcs, torch.Tensor) else list(srcs))
        cls._keep.extend(t for t in (wpk, kw.get("bias"), kw.get("mask"), kw.get("res0"), kw.get("res1")) if t is not None)
        half = n // 2
        for k, rng in enumerate(((0, half), (half, n))):
            with torch.cuda.stream(cls._stream(k)):
                K.conv3x3(srcs, wpk, cout, out=out, images=rng, strips=2 if k else True, plain_stores=(cls.plain_stores == "all" or (forward and cls.plain_stores == "fwd")), **kw)
        return out

    @classmethod
    def join(cls):
        try:
            if cls._forked:
                cur = torch.cuda.current_stream()
                for k in range(2):
                    if cls._stream(k) != cur:
                        cur.wait_stream(cls._stream(k))
        finally:
            # (also when a wait raises -- a stream capture that died mid-chain: the next step must fork afresh)
            cls._forked = False
            cls._main = None
            cls._keep.clear()

    @classmethod
    def reset(cls):
        """Forget a fork without waiting (after a failed capture + a device synchronisation)."""
        cls._forked = False
        cls._main = None
        cls._keep.clear()

    @classmethod
    def end_of_node(cls, backward):
        """Called by an autograd node that used the chains, before it hands tensors back."""
        if not (cls.lazy_bwd if backward else cls.lazy_fwd):
            cls.join()


class PackedConv:
    """Kernel-layout images of one conv weight in persistent device buffers.

    Staleness: in-place optimizer kernels (torch's fused AdamW) do not bump Tensor._version, so
    while gradients are enabled the images are rebuilt for every forward -- by ONE batched launch
    for the whole network (pack_all) when the owner calls it first, else per conv.  Without
    gradients (inference) they are cached until the weight's storage/version moves or
    invalidate() is called (restore / load_state_dict / optimizer step do)."""

    __slots__ = ("weight", "bias", "cin_pad", "slices", "_key", "_packs", "_bufs", "_prepacked", "grad_inplace",
                 "joint", "joint_slot")

    def __init__(self, weight, bias, cin_pad=None, slices=None):
        self.weight, self.bias, self.cin_pad = weight, bias, cin_pad
        # True once a GradBucket owns weight.grad / bias.grad: the wgrad kernels then write the
        # gradients there directly and the autograd nodes return None for them.
        self.grad_inplace = False
        self.slices = slices  # list of (cin_off, cin) for a conv over concatenated inputs, or None
        self._key = None
        self._packs = None
        self._bufs = None
        self._prepacked = False
        self.joint = None       # JointBwd: this conv's dgrad image lives in a shared arena (see there)
        self.joint_slot = 0

    def invalidate(self):
        self._key = None
        self._prepacked = False

    def _current_key(self):
        w = self.weight
        return (w.data_ptr(), w._version, str(w.device))

    def jobs(self):
        """Pack jobs (w, fwd_buf, bwd_buf, cout, cin_k, cin_off) writing into persistent buffers."""
        w = self.weight
        cout, cin_total = int(w.shape[0]), int(w.shape[1])
        dev = w.device
        if self._bufs is None or self._bufs[0] != (w.data_ptr(), str(dev)):
            def buf(a, b):
                return torch.empty(K.packed_weight_floats(a, b), device=dev, dtype=torch.float32)
            if self.slices is None:
                cin_k = cin_total if self.cin_pad is None else self.cin_pad
                bwd = None if self.cin_pad is not None else \
                    (self.joint.view(self.joint_slot, dev) if self.joint is not None else buf(cin_k, cout))
                specs = [(buf(cout, cin_k), bwd, cin_k, 0)]
            else:
                specs = [(buf(cout, cin_total), None, cin_total, 0)]
                specs += [(buf(cout, c), buf(c, cout), c, o) for (o, c) in self.slices]
            self._bufs = ((w.data_ptr(), str(dev)), specs)
        wd = w.detach()
        return [(wd, f, b, cout, cin_k, off) for (f, b, cin_k, off) in self._bufs[1]]

    def _mark_packed(self):
        self._packs = [(f, b) for (f, b, _, _) in self._bufs[1]]
        self._key = self._current_key()

    def refresh(self):
        """Called by the owning module's forward (outside autograd.Function, where the grad mode
        is visible)."""
        if self._prepacked:  # packed by pack_all() for exactly this use
            self._prepacked = False
            return self
        if self._key != self._current_key() or (torch.is_grad_enabled() and self.weight.requires_grad):
            with torch.no_grad():
                K.pack_weights_batch(self.jobs())
            self._mark_packed()
        return self

    def get(self):
        """-> list of (wpk_fwd, wpk_bwd) per slice (a single entry when slices is None)."""
        if self._packs is None:
            self.refresh()
        return self._packs


class JointBwd:
    """Two convolutions that read the SAME tensor (the first conv of body i+1 and the first conv of
    exit i's leg both read fea_i): the gradient of that tensor is the sum of their two input
    gradients, i.e. ONE convolution over the channel concatenation of their output gradients with
    the two dgrad weight images stacked along K.  The packed dgrad images are chunk-major, so
    stacking is adjacency: both PackedConvs write their image into one arena and the arena IS the
    stacked image.  Replaces two conv launches + autograd's accumulation add by one launch."""

    def __init__(self, first, second):
        cout, cin = int(first.weight.shape[0]), int(first.weight.shape[1])
        if tuple(second.weight.shape) != tuple(first.weight.shape) or first.slices or second.slices:
            raise RuntimeError("larvanet_amd: joint input gradients need two plain convs of one shape")
        self.floats = K.packed_weight_floats(cin, cout)
        self.cin = cin
        self._arena = None
        first.joint, first.joint_slot = self, 0
        second.joint, second.joint_slot = self, 1

    def arena(self, dev):
        if self._arena is None or self._arena.device != dev:
            self._arena = torch.empty(2 * self.floats, device=dev, dtype=torch.float32)
        return self._arena

    def view(self, slot, dev):
        return self.arena(dev)[slot * self.floats:(slot + 1) * self.floats]


class JointInputGrad:
    """Book-keeping of the fusion above inside one StepScope.  Backward: the exits run first (one
    node for all of them, ExitsFn); for every exit whose leg shares its input with the next body's
    first conv they leave the leg's last dgrad launch undone and park its operand here; the body,
    when its own backward reaches its first conv, issues the one stacked convolution and returns
    the complete gradient of the shared tensor."""

    active = False
    _parked = {}   # data_ptr of the shared tensor -> (dh of the leg's first conv, its PackedConv)

    @classmethod
    def reset(cls):
        cls._parked = {}

    @classmethod
    def can_park(cls, pc):
        return cls.active and pc.joint is not None and pc.joint_slot == 1

    @classmethod
    def park(cls, fea, dh, pc):
        cls._parked[fea.data_ptr()] = (dh, pc)

    @classmethod
    def take(cls, x, pc):
        """dh parked by the exit that reads `x`, if `pc` (a body's first conv) is its arena partner."""
        if not cls._parked:
            return None
        hit = cls._parked.get(x.data_ptr())
        if hit is None or pc.joint is None or hit[1].joint is not pc.joint or pc.joint_slot != 0:
            return None
        del cls._parked[x.data_ptr()]
        return hit[0]


def step_prologue(pcs, x):
    """One launch for what precedes a training step's first conv: every conv's packed images (pack_all),
    the head's 16-channel padded input and the bicubic x4 base image -> (x16, base), or None when the
    shape / job count does not fit that launch (the caller then issues the three steps separately)."""
    jobs = []
    for pc in pcs:
        jobs += pc.jobs()
    N, C, H, W = (int(v) for v in x.shape)
    if len(jobs) > 64 or C > 16 or PaddedWidth.current is not None or not x.is_contiguous():
        return None
    x16 = StepScope.padded_input((N, 16, H, W), x.device)
    base = torch.empty((N, C, 4 * H, 4 * W), device=x.device, dtype=torch.float32)
    with torch.no_grad():
        K.step_prologue(jobs, x.detach(), x16, base)
    for pc in pcs:
        pc._mark_packed()
        pc._prepacked = True
    return x16, base


def pack_all(pcs):
    """One launch (per 64 jobs) packing every conv of a network; each PackedConv then skips its
    own repack at its next refresh()."""
    jobs = []
    for pc in pcs:
        jobs += pc.jobs()
    with torch.no_grad():
        K.pack_weights_batch(jobs)
    for pc in pcs:
        pc._mark_packed()
        pc._prepacked = True


class GradBucket:
    """All parameter gradients of a network as views of ONE flat fp32 buffer: the weight-gradient
    kernels write straight into it, the data-parallel all-reduce is a single collective over it
    (no flatten / unflatten copies), and the optimizer sees ordinary p.grad tensors.
    Gradients are OVERWRITTEN by every backward (the reference zeroes them before each backward,
    models/LarvaNet.py:112-113, so the result is the same); zero_grad(set_to_none=True) would
    detach the views and must not be used while a bucket is attached."""

    def __init__(self, module, pcs):
        params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self._pairs = []
        for p in params:
            view = self.flat[off:off + p.numel()].view_as(p)
            p.grad = view
            self._pairs.append((p, view))
            off += p.numel()
        owned = {id(p) for p in params}
        for pc in pcs:
            pc.grad_inplace = id(pc.weight) in owned and id(pc.bias) in owned

    def intact(self, module=None):
        """False if somebody replaced or dropped a .grad (then fall back to autograd's own)."""
        return all(p.grad is view for p, view in self._pairs)

    def span(self, tensors):
        """[lo, hi) in floats of the part of the flat buffer that `tensors` (views of it) cover."""
        base = self.flat.data_ptr()
        lo, hi = self.flat.numel(), 0
        for t in tensors:
            off = (t.data_ptr() - base) // 4
            if off < 0 or off + t.numel() > self.flat.numel():
                return None
            lo, hi = min(lo, off), max(hi, off + t.numel())
        return (lo, hi) if hi > lo else None


def _targets(pc):
    """(dw, db) destinations of a conv's gradients: the bucket views, or None = allocate."""
    if pc is not None and pc.grad_inplace and pc.weight.grad is not None and pc.bias.grad is not None:
        return pc.weight.grad, pc.bias.grad
    return None, None


def _splits(njobs, cout=48, cin=48):
    """Workgroups per layer of a weight-gradient launch over `njobs` layers: one workgroup per CU in total, two where
    two of that shape's workgroups fit a CU (kernels.wgrad_cu_share)."""
    return max(1, _WGRAD_WORKGROUPS * K.wgrad_cu_share(cout, cin) // njobs)


class DeferredWgrad:
    """Weight gradients are never on the critical path of backward, and their cost has a part
    that scales with the number of WORKGROUPS launched, not with the work: every workgroup writes
    one partial image (83 KB at 48x48 channels) that the fixed-order reduction reads back.  One
    launch per autograd node (8 layers x 32 workgroups, 2 x 128 for a leg) means ~2000 partial
    images = 340 MB of extra traffic per step.  Inside `with StepScope(...)` the layers whose
    gradient is written in place (GradBucket) are therefore only QUEUED during backward and issued
    when the scope ends: all layers of one kernel shape in as few launches as the job table
    allows, each launch ~256 workgroups = one per CU.  Only in-place targets are deferred: a
    tensor handed back to autograd must be complete when backward() returns it."""

    active = False
    jobs_per_launch = int(os.environ.get("LARVA_WGRAD_JOBS", "32"))
    _pending = {}   # (cout, cin) -> list of jobs (they keep dy / x / targets alive)

    # (Round 4, measured and removed: the exits' eight layers issued BESIDE the backward chain as a small grid on a
    # stream of their own -- they are complete when the chain starts, and a weight-gradient workgroup owns a whole CU.  The
    # workgroups did get their CUs within 16 us, but every chain launch is 256 workgroups for 512 slots: with 64 CUs gone
    # the two chains queue for slots, fall into step and run at 18.6 instead of 14.6 us per layer -- step 1.78 against
    # 1.67 ms with 48-128 early workgroups, 1.86 with 32.  profiles/r04_ab_early_wgrad.txt, r04_step_timeline_early64.txt.)
    @classmethod
    def push(cls, cout, cin, jobs):
        cls._pending.setdefault((cout, cin), []).extend(jobs)

    _late = []      # launches held back by a split flush: [(cout, cin, jobs)]

    # one grid over all layers of a launch (kernels.conv3x3_wgrad_partial_flat) where the shape allows
    flat = os.environ.get("LARVA_WGRAD_FLAT", "1") != "0"
    head_in_flat = os.environ.get("LARVA_WGRAD_HEAD_IN_FLAT", "1") != "0"

    @classmethod
    def _launches(cls, split=False):
        pending, cls._pending = cls._pending, {}
        cap = max(1, min(cls.jobs_per_launch, K.max_wgrad_jobs()))
        if cls.flat and not split:
            cap = K.max_wgrad_jobs()   # a flat grid has no preferred layer count: as many layers per launch as fit
        out = []
        for (cout, cin), jobs in pending.items():
            # from the end of the gradient bucket downwards (= roughly the order backward produced
            # them): a split flush then completes a contiguous suffix of the bucket first
            jobs = sorted(jobs, key=lambda j: -j["dw"].data_ptr())
            # full launches of `cap` layers, then the rest: 40 layers -> 32 x 8 workgroups + 8 x 32
            # workgroups, both exactly one workgroup per CU (20 + 20 would leave 16 CUs idle twice)
            out += [(cout, cin, jobs[i:i + cap]) for i in range(0, len(jobs), cap)]
        out.sort(key=lambda l: -max(j["dw"].data_ptr() for j in l[2]))
        return out

    # the step's loss (mean of the exits' terms), left to ride on the last reduction launch of backward:
    # (terms, scales, divisor, out) or None; see MeanTermsFn
    pending_loss = None

    @classmethod
    def finish_loss(cls):
        """The loss nobody has finished yet (no reduction launch took it along): its own launch."""
        job, cls.pending_loss = cls.pending_loss, None
        if job is not None:
            terms, scales, divisor, out = job
            out.copy_(K.loss_from_partials(terms, scales, divisor))

    @staticmethod
    def _issue(launches):
        """The partial-image launches one after the other, then ONE fixed-order reduction over all
        of them (each layer with its own split count and kernel shape)."""
        reduce_jobs = []
        # the 3 -> 48 head (one (48, 16) layer on its padded input) rides at the end of the last flat (48, 48) grid
        # of the same image geometry instead of having a launch of its own
        head = host = None
        if DeferredWgrad.flat and DeferredWgrad.head_in_flat:
            heads = [l for l in launches if (l[0], l[1]) == (48, 16) and len(l[2]) == 1]
            hosts = [l for l in launches if (l[0], l[1]) == (48, 48)]
            if len(heads) == 1 and hosts and hosts[-1][2][0]["dy"].shape == heads[0][2][0]["dy"].shape:
                head, host = heads[0], hosts[-1]
        todo = [l for l in launches if l is not head]
        for launch in todo:   # (may grow: a head whose host grid did not apply goes its own way at the end)
            cout, cin, chunk = launch
            extra = head[2][0] if launch is host else None
            res = K.conv3x3_wgrad_partial_flat(chunk, cout, cin, _WGRAD_WORKGROUPS, head=extra) if DeferredWgrad.flat else None
            if res is None and extra is not None:
                todo.append(head)
                extra = None
            if res is not None:
                reduce_jobs += [dict(j, partial=p, splits=s, cout=cout, cin=cin) for j, p, s in zip(chunk, *res)]
                if extra is not None:
                    reduce_jobs.append(dict(extra, partial=res[0][-1], splits=res[1][-1], cout=48, cin=16))
                continue
            parts, used = K.conv3x3_wgrad_partial(chunk, cout, cin, _splits(len(chunk), cout, cin))
            reduce_jobs += [dict(j, partial=p, splits=used, cout=cout, cin=cin) for j, p in zip(chunk, parts)]
        for i in range(0, len(reduce_jobs), 64):
            last = i + 64 >= len(reduce_jobs)
            loss, DeferredWgrad.pending_loss = (DeferredWgrad.pending_loss, None) if last else (None, DeferredWgrad.pending_loss)
            K.wgrad_reduce(reduce_jobs[i:i + 64], loss=loss)

    @classmethod
    def flush(cls, split=False):
        """Issue the queued layers.  split: only the first half of the launches (the layers
        backward reached first = the END of the flat gradient bucket) goes out now; the rest
        waits for flush_late(), so that a data-parallel caller can start all-reducing the first
        half while the second is still being computed.  Returns the tensors the issued launches
        write (dw, db ...) when split, else None."""
        launches = cls._launches(split)
        n_early = max(1, len(launches) // 2) if split else len(launches)
        cls._issue(launches[:n_early])
        cls._late = launches[n_early:]
        if not split:
            return None
        return [t for _, _, chunk in launches[:n_early] for j in chunk for t in (j["dw"], j["db"]) if t is not None]

    @classmethod
    def late_targets(cls):
        return [t for _, _, chunk in cls._late for j in chunk for t in (j["dw"], j["db"]) if t is not None]

    @classmethod
    def flush_late(cls):
        late, cls._late = cls._late, []
        cls._issue(late)

    @classmethod
    def drop(cls):
        cls._pending = {}
        cls._late = []
        cls.pending_loss = None


class StepScope:
    """The plugin's forward+backward of one batch: optional side streams, deferred wgrad.
    Leaving the scope joins the side streams and issues the queued weight gradients, so they are
    complete on the current stream afterwards (also as the tail of a hipGraph capture)."""

    depth = 0
    _padded = {}
    # Value of the gradient that loss.backward() will be seeded with, when the caller of the scope
    # guarantees it (the plugin seeds with 1 and backpropagates the mean of the exits untouched):
    # the exits then write their L1 gradient during the FORWARD pass, in the same sweep over
    # (output, truth) that computes the loss.  None = unknown, the exits do it in backward.
    seed_grad = None
    # The caller reads the loss before backward has run (the plugin's early-loss captures): MeanTermsFn then finishes
    # it at once instead of leaving it to the last reduction launch of backward -- True, or the kernels.HostCell
    # the finishing launch stores the value into as well.
    early_loss = False

    @classmethod
    def padded_input(cls, shape, device):
        """Zero tensor for the head's channel-padded input.  Inside a scope it is a cached buffer
        (zeroed once; the caller overwrites the same sub-block every step), elsewhere a fresh one."""
        if cls.depth == 0:
            return torch.zeros(shape, device=device, dtype=torch.float32)
        key = (tuple(shape), str(device))
        buf = cls._padded.get(key)
        if buf is None:
            # Never evicted: captured graphs (the training step, up to four inference shapes per model) bake this
            # buffer's address in, and a freed buffer's memory would be handed to somebody else.  Scoped shapes are
            # the training batch shapes and the captured inference shapes -- a handful of N x 16 x H x W buffers.
            buf = cls._padded[key] = torch.zeros(shape, device=device, dtype=torch.float32)
        return buf

    def __init__(self, side_streams=False, defer_wgrad=True, split_flush=False, joint_input_grads=True,
                 seed_grad=None, dual_chain=False, lazy_chain_joins=(False, False), early_loss=False):
        self.seed_grad_value = seed_grad
        self.early_loss_value = early_loss or False
        self.dual_chain = dual_chain
        self.lazy_chain_joins = lazy_chain_joins
        self.side_streams = side_streams
        self.defer_wgrad = defer_wgrad
        self.split_flush = split_flush
        self.joint_input_grads = joint_input_grads
        self.early_targets = None   # split flush: tensors complete when the scope ends

    def __enter__(self):
        gpu = torch.cuda.is_available()
        SideStreams.active = bool(self.side_streams) and gpu
        DeferredWgrad.active = bool(self.defer_wgrad) and gpu
        DeferredWgrad.drop()
        # (with side streams the exit and the next body run concurrently: keep them independent)
        JointInputGrad.active = bool(self.joint_input_grads) and gpu and not SideStreams.active
        JointInputGrad.reset()
        DualChain.enabled = bool(self.dual_chain) and gpu and not SideStreams.active
        DualChain.lazy_fwd, DualChain.lazy_bwd = (bool(v) and DualChain.enabled for v in self.lazy_chain_joins)
        StepScope.depth += 1
        StepScope.seed_grad = self.seed_grad_value
        StepScope.early_loss = self.early_loss_value
        return self

    def __exit__(self, exc_type, *exc):
        try:
            DualChain.join()   # the dgrad chains end here; the queued weight gradients read what they wrote
            if SideStreams.active:
                SideStreams.join()
            if exc_type is None:
                self.early_targets = DeferredWgrad.flush(split=self.split_flush)
                DeferredWgrad.finish_loss()
            else:
                DeferredWgrad.drop()
            if exc_type is None and JointInputGrad._parked:
                raise RuntimeError("larvanet_amd: an exit parked its input gradient but the next body never ran backward")
        finally:
            DualChain.reset()
            StepScope.depth -= 1
            StepScope.seed_grad = None
            StepScope.early_loss = False
            DeferredWgrad._pending = {}
            SideStreams.active = False
            DualChain.enabled = DualChain.lazy_fwd = DualChain.lazy_bwd = False
            DeferredWgrad.active = False
            JointInputGrad.active = False
            JointInputGrad.reset()
        return False


def _wgrad(jobs, cout, cin, inplace=False):
    """jobs: list of (dy, x, weight shape, cin_off, cin_valid, shared dw or None[, shared db]) ->
    list of (dw, db).  inplace: every dw/db that matters is a GradBucket view, so the whole
    job may be deferred to the end of the StepScope.  Inside a SideStreams scope with
    wgrad_on_side the launches go to the wgrad side stream."""
    side = None
    if SideStreams.active and SideStreams.wgrad_on_side:
        side = SideStreams.fork("wgrad", *[t for j in jobs for t in (j[0], j[1])])
    defer = inplace and DeferredWgrad.active and side is None
    jobs = list(jobs)
    ctx = torch.cuda.stream(side) if side is not None else _NullCtx()
    out, batch = [], []
    with ctx:
        for job in jobs:
            (dy, x, wshape, cin_off, cin_valid, dw_shared) = job[:6]
            db_shared = job[6] if len(job) > 6 else None
            dw = dw_shared if dw_shared is not None else torch.empty(wshape, device=dy.device, dtype=torch.float32)
            if db_shared is not None or not defer:
                db = db_shared if db_shared is not None else torch.empty((cout,), device=dy.device,
                                                                         dtype=torch.float32)
            else:
                db = None  # in-place mode and nobody wants this bias gradient
            batch.append({"dy": dy, "x": x, "dw": dw, "db": db, "cin_off": cin_off, "cin_valid": cin_valid})
            out.append((dw, db))
        if defer:
            DeferredWgrad.push(cout, cin, batch)
            batch = []
        for i in range(0, len(batch), 16):
            chunk = batch[i:i + 16]
            parts = K.conv3x3_wgrad(chunk, cout, cin, _splits(len(chunk), cout, cin))
            if side is not None:
                SideStreams.keep(*parts)
    return out


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# ONE rule for "a large inference batch" (a whole validation image rather than a batch of patches): above it the
# inference forward runs as eager launches (models/LarvaNet.py `_infer`: 36 launches of ~60 us each, the host is
# milliseconds ahead) and the head takes the direct K = 27 kernel (33 MB of output: HBM-bound).  Both decisions use
# the same count -- N x H x W LR pixels of the INPUT, unpadded -- so that no shape takes one without the other.
LARGE_INFERENCE_PIXELS = 100000


def is_large_inference(n, h, w):
    return int(n) * int(h) * int(w) > LARGE_INFERENCE_PIXELS


def _head_direct_setting():
    """LARVA_HEAD_DIRECT: auto (default) | 0 = never the direct head kernel | 1 = always."""
    v = os.environ.get("LARVA_HEAD_DIRECT", "auto")
    if v not in ("auto", "0", "1"):
        raise RuntimeError("larvanet_amd: LARVA_HEAD_DIRECT=%r (auto, 0 or 1)" % v)
    return {"0": False, "1": True}.get(v, "auto")


class HeadFn(torch.autograd.Function):
    """LarvaHead.forward (models/LarvaNet.py:223-233): conv3x3 3->48, no activation.
    Forward: the MFMA conv kernel on the image zero-padded to two 8-channel K chunks (13/16 of its
    multiplies are zeros, still the faster one) or, LARVA_HEAD_DIRECT=1, the direct K = 27 kernel on
    the raw 3-channel image (larva_head_conv3_direct; rocprof A/B in profiles/).
    The weight gradient (M = 48, N = 27, K = all pixels: a real GEMM) stays on the MFMA wgrad kernel,
    which reads the 16-channel padded image -- so that copy is made whenever a backward will follow."""

    # rocprofv3 A/B at 16 x 3 x 48 x 48 (profiles/README.md, r02_head_*): padded-MFMA launch 7.2 us, direct
    # kernel 9.0 us (LDS-broadcast weights; 12.8 us with scalar-loaded weights) -> MFMA at the training size.  A whole
    # validation image is another matter: 33 MB of output, the padded MFMA launch 25-31 us against 15 us for the direct
    # kernel's 4-pixel x 8-channel threads with 16-byte stores (profiles/r06_head_bicubic_ab.txt): "auto" = direct for
    # inference on more than LARGE_INFERENCE_PIXELS LR pixels (is_large_inference); 0 / 1 = never / always.
    direct = _head_direct_setting()

    @staticmethod
    def forward(ctx, x, weight, bias, pc, x16=None, training=True):
        """training: a backward may follow (the caller's view: inside forward() the grad mode is always off and
        needs_input_grad reports the parameters' requires_grad even under no_grad)."""
        N, C, H, W = x.shape
        P = PaddedWidth.pitch_of(W) if _lw() is not None else W
        cout = int(weight.shape[0])
        training = bool(training) and bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        want = HeadFn.direct if HeadFn.direct != "auto" else (not training and is_large_inference(N, H, W))
        use_direct = want and C == 3 and cout % 16 == 0
        if x16 is not None:      # prepared by the step's prologue launch (step_prologue)
            pass
        elif training or not use_direct:
            x16 = StepScope.padded_input((N, 16, H, P), x.device)  # channels C..15 / columns W..P-1 stay zero
            x16[:, :C, :, :W] = x
        if use_direct:
            out = K.head_conv3_direct(x, weight.detach(), bias.detach(), pitch=P)
        else:
            (fwd, _), = pc.get()
            out = DualChain.conv(x16, fwd, cout, forward=True, bias=bias.detach(), logical_w=_lw())
            DualChain.end_of_node(False)
        if x16 is not None:
            ctx.save_for_backward(x16)
        ctx.wshape = tuple(weight.shape)
        ctx.pc = pc
        return out

    @staticmethod
    def backward(ctx, dy):
        (x16,) = ctx.saved_tensors
        dy = dy.contiguous()
        cout, cin = ctx.wshape[0], ctx.wshape[1]
        tw, tb = _targets(ctx.pc)
        if not (tw is not None and DeferredWgrad.active):
            DualChain.join()   # dy comes off the two dgrad chains and is read right here
        (dw, db), = _wgrad([(dy, x16, ctx.wshape, 0, cin, tw, tb)], cout, 16, inplace=tw is not None)
        if tw is not None:
            return None, None, None, None, None, None
        return None, dw, db, None, None, None


class BodyFn(torch.autograd.Function):
    """LarvaBody.forward (models/LarvaNet.py:236-248): x + res_blocks(x), each block
    x + conv2(relu(conv1(x))) (models/LarvaNet.py:205-220).  2 launches per block forward."""

    @staticmethod
    def forward(ctx, x, pcs, *params):
        # params = (w1, b1, w2, b2) per block; pcs = [PackedConv] in the same conv order
        nb = len(params) // 4
        keep = [x]
        fea = x
        for j in range(nb):
            w1, b1, w2, b2 = params[4 * j:4 * j + 4]
            (f1, _), = pcs[2 * j].get()
            (f2, _), = pcs[2 * j + 1].get()
            c = int(w1.shape[0])
            h = DualChain.conv(fea, f1, c, forward=True, bias=b1.detach(), relu=True, logical_w=_lw())
            if j == nb - 1:
                nxt = DualChain.conv(h, f2, c, forward=True, bias=b2.detach(), res0=fea, res1=x, logical_w=_lw())
            else:
                nxt = DualChain.conv(h, f2, c, forward=True, bias=b2.detach(), res0=fea, logical_w=_lw())
            keep.append(h)
            if j < nb - 1:
                keep.append(nxt)
            fea = nxt
        DualChain.end_of_node(False)
        ctx.save_for_backward(*keep)
        ctx.pcs = pcs
        ctx.nb = nb
        ctx.wshape = tuple(params[0].shape)
        return fea

    @staticmethod
    def backward(ctx, dy):
        nb, pcs = ctx.nb, ctx.pcs
        keep = ctx.saved_tensors[:2 * nb]
        dy = dy.contiguous()
        c = ctx.wshape[0]
        # keep = [x, h0, fea1, h1, fea2, ..., h_{nb-1}]
        g = dy
        jobs = [None] * (2 * nb)
        dx = None
        for j in reversed(range(nb)):
            fea_j = keep[0] if j == 0 else keep[2 * j]
            h_j = keep[2 * j + 1]
            (_, bw1), = pcs[2 * j].get()
            (_, bw2), = pcs[2 * j + 1].get()
            dh = DualChain.conv(g, bw2, c, mask=h_j)
            jobs[2 * j + 1] = (g, h_j, ctx.wshape, 0, c) + _targets(pcs[2 * j + 1])
            jobs[2 * j] = (dh, fea_j, ctx.wshape, 0, c) + _targets(pcs[2 * j])
            if j > 0:
                g = DualChain.conv(dh, bw1, c, res0=g)
            else:
                dh_leg = JointInputGrad.take(keep[0], pcs[0])
                if dh_leg is None:
                    dx = DualChain.conv(dh, bw1, c, res0=g, res1=dy)
                else:
                    # x is also read by the previous exit's leg, which left its last dgrad to us:
                    # d x = dgrad(conv1)(dh) + g + dy + dgrad(leg conv1)(dh_leg) as ONE launch over
                    # K = [dh ; dh_leg] with the two dgrad images stacked (JointBwd arena)
                    dx = DualChain.conv([dh, dh_leg], pcs[0].joint.arena(dh.device), c, res0=g, res1=dy)
        inplace = all(_targets(pc)[0] is not None for pc in pcs)
        if not (inplace and DeferredWgrad.active):
            DualChain.join()       # the weight gradients are computed right here, on this stream
        DualChain.end_of_node(True)
        grads = _wgrad(jobs, c, c, inplace=inplace)
        flat = []
        for pc, (dw, db) in zip(pcs, grads):
            flat += [None, None] if _targets(pc)[0] is not None else [dw, db]
        return (dx, None) + tuple(flat)


class LegFn(torch.autograd.Function):
    """LarvaLeg.forward (models/LarvaNet.py:251-267): conv+ReLU, conv -> PixelShuffle(4) -> += base,
    the shuffle and the base add fused into the second conv's store."""

    @staticmethod
    def forward(ctx, fea, base, pcs, w1, b1, w2, b2):
        (f1, _), = pcs[0].get()
        (f2, _), = pcs[1].get()
        c = int(w1.shape[0])
        h = K.conv3x3(fea, f1, c, bias=b1.detach(), relu=True, logical_w=_lw())
        out = K.conv3x3(h, f2, int(w2.shape[0]), bias=b2.detach(), shuffle=True, base=base, logical_w=_lw())
        ctx.save_for_backward(fea, h)
        ctx.pcs = pcs
        ctx.wshape, ctx.wshape2 = tuple(w1.shape), tuple(w2.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        fea, h = ctx.saved_tensors[:2]
        pcs = ctx.pcs
        c, c2 = ctx.wshape[0], ctx.wshape2[0]
        (_, bw1), = pcs[0].get()
        (_, bw2), = pcs[1].get()
        dyl = K.pixel_unshuffle4(dout.contiguous())
        dh = K.conv3x3(dyl, bw2, c, mask=h)
        dfea = K.conv3x3(dh, bw1, c)
        ((dw1, db1),), ((dw2, db2),) = _leg_wgrad([(dh, fea, ctx.wshape, 0, c) + _targets(pcs[0])],
                                                    [(dyl, h, ctx.wshape2, 0, c) + _targets(pcs[1])], c, c2)
        if _targets(pcs[0])[0] is not None:
            dw1 = db1 = None
        if _targets(pcs[1])[0] is not None:
            dw2 = db2 = None
        # base comes from a parameter-free interpolation of the network input: no gradient
        return dfea, None, None, dw1, db1, dw2, db2


def _leg_wgrad(first, second, c, c2):
    """Weight gradients of legs' first (c -> c) and last (c -> c2) convs, jobs as for _wgrad -> (results of `first`,
    results of `second`).  One call when the two shapes agree (c2 == c: the reference's 48-channel network), else one
    per shape (--num_filters 32 / 64: the last conv keeps 48 = 3 * 4**2 outputs)."""
    inplace = all(j[5] is not None for j in list(first) + list(second))
    if c2 == c:
        res = _wgrad([j for pair in zip(first, second) for j in pair], c, c, inplace=inplace)
        return res[0::2], res[1::2]
    return _wgrad(list(first), c, c, inplace=inplace), _wgrad(list(second), c2, c, inplace=inplace)


class ExitFn(torch.autograd.Function):
    """One whole exit of the training step: LarvaLeg.forward followed by nn.L1Loss against the
    truth (models/LarvaNet.py:107-108).  Same kernels as LegFn + L1LossFn, but the backward
    writes the L1 gradient directly in the pixel-unshuffled layout the leg's dgrad/wgrad read
    (one fused launch instead of l1_bwd + pixel_unshuffle, no HR-layout gradient tensor).
    Returns (exit image, loss term); the image output carries no gradient path of its own."""

    @staticmethod
    def forward(ctx, fea, base, truth, pcs, w1, b1, w2, b2, divisor=None):
        (f1, _), = pcs[0].get()
        (f2, _), = pcs[1].get()
        c = int(w1.shape[0])
        h = K.conv3x3(fea, f1, c, bias=b1.detach(), relu=True)
        out = K.conv3x3(h, f2, int(w2.shape[0]), bias=b2.detach(), shuffle=True, base=base)
        dyl = None
        if divisor is None:
            term = K.l1_fwd(out, truth)   # the finished L1 value
            ctx.gscale = 1.0
        else:
            # the term as it enters the mean over `divisor` exits: block partial sums of |out-truth|,
            # finished by MeanTermsFn together with the other exits (see LossTerm); its gradient
            # arrives unscaled and the 1/divisor is applied inside the L1 backward kernel
            ctx.gscale = float(np.float32(1.0) / np.float32(divisor))
            if StepScope.seed_grad is not None:  # gradient value known now: one sweep does both
                term, _, dyl = K.l1_partial_grad(out, truth, StepScope.seed_grad, ctx.gscale)
            else:
                term, _ = K.l1_partial(out, truth)
        ctx.have_dyl = dyl is not None
        if dyl is not None:
            ctx.save_for_backward(fea, h, dyl)
        else:
            ctx.save_for_backward(fea, h, out, truth)
        ctx.pcs = pcs
        ctx.wshape, ctx.wshape2 = tuple(w1.shape), tuple(w2.shape)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)  # no 7 MB zero gradient for the non-differentiable image output
        return out, term

    @staticmethod
    def backward(ctx, _dout, gterm):
        if ctx.have_dyl:
            fea, h, dyl = ctx.saved_tensors[:3]
        else:
            fea, h, out, truth = ctx.saved_tensors[:4]
        pcs = ctx.pcs
        c = ctx.wshape[0]
        if gterm is None:
            return (None,) * 9
        (_, bw1), = pcs[0].get()
        (_, bw2), = pcs[1].get()
        if not ctx.have_dyl:
            # (a partial-sum term receives its scalar gradient broadcast to its shape: element 0)
            g0 = gterm.as_strided((), ()) if gterm.dim() else gterm.contiguous()
            dyl = K.l1_bwd_unshuffle4(out, truth, g0, ctx.gscale)
        dh = K.conv3x3(dyl, bw2, c, mask=h)
        dfea = K.conv3x3(dh, bw1, c)
        ((dw1, db1),), ((dw2, db2),) = _leg_wgrad([(dh, fea, ctx.wshape, 0, c) + _targets(pcs[0])],
                                                    [(dyl, h, ctx.wshape2, 0, c) + _targets(pcs[1])], c, ctx.wshape2[0])
        if _targets(pcs[0])[0] is not None:
            dw1 = db1 = None
        if _targets(pcs[1])[0] is not None:
            dw2 = db2 = None
        return dfea, None, None, None, dw1, db1, dw2, db2, None


def _conv_group(jobs, cout, **kw):
    """Independent same-shape convs: batched launches of up to 4 jobs, a lone one on its own."""
    outs = []
    for i in range(0, len(jobs), 4):
        chunk = jobs[i:i + 4]
        if len(chunk) == 1:
            j = chunk[0]
            outs.append(K.conv3x3(j["srcs"], j["wpk"], cout, bias=j.get("bias"), mask=j.get("mask"), base=j.get("base"),
                                  res0=j.get("res0"), res1=j.get("res1"), **kw))
        else:
            outs += K.conv3x3_batch(chunk, cout, **kw)
    return outs


class ExitsFn(torch.autograd.Function):
    """ALL exits of the training step as one autograd node (models/LarvaNet.py:104-108 for every i):
    exit i = LarvaLeg(fea_i, base) scored by nn.L1Loss against the truth.  The exits do not depend
    on each other, and a conv launch at the training shape leaves half of every CU's time unused
    (one workgroup per CU, the kernel fits two): here the M first convs are ONE batched launch, the
    M pixel-shuffle convs another, and in backward the M mask-dgrads a third (4 jobs: 55 us
    against 67 us one by one).  Per exit the node returns the block partial sums of its L1 term
    (LossTerm, prescaled by 1/divisor in its gradient) and, once, the LAST exit's image.
    Backward leaves the input gradient of every exit whose leg shares its input with the next
    body's first conv to that body (JointInputGrad)."""

    fuse_l1 = os.environ.get("LARVA_FUSE_EXIT_L1", "1") != "0"   # L1 inside the pixel-shuffle conv launch

    @staticmethod
    def forward(ctx, base, truth, legs, divisor, *args):
        M = len(legs)  # legs: [[PackedConv conv1, PackedConv conv2]] per exit
        feas, params = args[:M], args[M:]   # params: (w1, b1, w2, b2) per exit
        c = int(params[0].shape[0])
        c2 = int(params[2].shape[0])
        hs = _conv_group([{"srcs": feas[i], "wpk": legs[i][0].get()[0][0], "bias": params[4 * i + 1].detach()}
                          for i in range(M)], c, relu=True)
        ctx.gscale = float(np.float32(1.0) / np.float32(divisor))
        ctx.have_dyl = StepScope.seed_grad is not None
        shuffle_jobs = [{"srcs": hs[i], "wpk": legs[i][1].get()[0][0], "bias": params[4 * i + 3].detach(), "base": base}
                        for i in range(M)]
        outs, parts, third = [None] * M, [None] * M, [None] * M
        fused = ctx.have_dyl and ExitsFn.fuse_l1
        if fused:
            # gradient value known now: every exit is scored inside its pixel-shuffle conv launch (partial
            # sums of |out - truth| and the sign gradient straight from the accumulators); only the last
            # exit's image is stored at all
            i = 0
            while i < M and fused:
                n = 3 if M - i == 5 else min(4, M - i)   # launches of 2..4 exits (5 left = 3 + 2)
                res = K.conv3x3_exit_l1_batch(shuffle_jobs[i:i + n], c2, truth, StepScope.seed_grad, ctx.gscale,
                                              [i + k == M - 1 for k in range(n)]) if n > 1 else None
                if res is None:
                    fused = False
                else:
                    outs[i:i + n], parts[i:i + n], third[i:i + n] = res
                    i += n
        if not fused:
            outs = _conv_group(shuffle_jobs, c2, shuffle=True)
            parts, third = [], []
            if ctx.have_dyl:  # one sweep over (out_i, truth) computes the partial sums and the gradient, all exits at once
                for i in range(0, M, 8):
                    p8, _, g8 = K.l1_partial_grad_batch(outs[i:i + 8], truth, StepScope.seed_grad, ctx.gscale)
                    parts += p8
                    third += g8
            else:
                for out in outs:
                    part, _ = K.l1_partial(out, truth)
                    parts.append(part)
                    third.append(out)
        ctx.save_for_backward(truth, *feas, *hs, *third)
        ctx.legs, ctx.M = legs, M
        ctx.wshape, ctx.wshape2 = tuple(params[0].shape), tuple(params[2].shape)
        ctx.mark_non_differentiable(outs[-1])
        ctx.set_materialize_grads(False)
        return (outs[-1],) + tuple(parts)

    @staticmethod
    def backward(ctx, _dout, *gterms):
        M, legs = ctx.M, ctx.legs
        saved = ctx.saved_tensors
        truth, feas, hs, third = saved[0], saved[1:1 + M], saved[1 + M:1 + 2 * M], saved[1 + 2 * M:1 + 3 * M]
        c = ctx.wshape[0]
        live = [i for i in range(M) if gterms[i] is not None]
        dyls = {}
        for i in live:
            if ctx.have_dyl:
                dyls[i] = third[i]
            else:
                g0 = gterms[i].as_strided((), ())  # the scalar gradient arrives broadcast to the term's shape
                dyls[i] = K.l1_bwd_unshuffle4(third[i], truth, g0, ctx.gscale)
        dhs = dict(zip(live, _conv_group([{"srcs": dyls[i], "wpk": legs[i][1].get()[0][1], "mask": hs[i]} for i in live], c))) if live else {}
        dfeas = [None] * M
        grads = [None] * (4 * M)
        jobs1, jobs2 = [], []
        for i in live:
            pc1, pc2 = legs[i]
            if JointInputGrad.can_park(pc1):
                JointInputGrad.park(feas[i], dhs[i], pc1)  # the next body adds it to its own input gradient
            else:
                # (the last exit: its input gradient is the first link of the backward layer chain)
                dfeas[i] = DualChain.conv(dhs[i], pc1.get()[0][1], c)
            jobs1.append((dhs[i], feas[i], ctx.wshape, 0, c) + _targets(pc1))
            jobs2.append((dyls[i], hs[i], ctx.wshape2, 0, c) + _targets(pc2))
        DualChain.end_of_node(True)
        for which, res in enumerate(_leg_wgrad(jobs1, jobs2, c, ctx.wshape2[0]) if jobs1 else ()):
            for i, (dw, db) in zip(live, res):
                if _targets(legs[i][which])[0] is None:
                    grads[4 * i + 2 * which], grads[4 * i + 2 * which + 1] = dw, db
        return (None, None, None, None) + tuple(dfeas) + tuple(grads)


class LossTerm:
    """What one exit contributes to the mean loss: a finished scalar (scale 1), or the block partial
    sums of sum|out - truth| with scale = 1 / numel whose producer (ExitFn with a divisor) applies
    the 1/n of the mean in its own backward kernel (`prescaled`)."""

    __slots__ = ("tensor", "scale", "prescaled")

    def __init__(self, tensor, scale=1.0, prescaled=False):
        self.tensor, self.scale, self.prescaled = tensor, float(scale), bool(prescaled)


class MeanTermsFn(torch.autograd.Function):
    """`loss += term` over the exits and `loss / num_modules` (models/LarvaNet.py:104-109) as one
    launch over scalars and/or partial sums (same arithmetic, same order as finishing every L1
    separately and adding the scalars).  A finished scalar receives the gradient g / n; a
    prescaled partial-sum term receives g itself (broadcast view, no kernel)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        ctx.meta = meta
        ctx.shapes = [tuple(t.shape) for t in tensors]
        ts, scales = [t.contiguous() for t in tensors], [m[0] for m in meta]
        if (len(ts) <= 8 and DeferredWgrad.active and StepScope.depth > 0 and not StepScope.early_loss
                and DeferredWgrad.pending_loss is None):
            # inside the plugin's step nobody reads the value before the step ends (backward is seeded with 1):
            # the finishing block rides on the weight-gradient reduction launch at the end of backward
            out = torch.empty((), device=ts[0].device, dtype=torch.float32)
            DeferredWgrad.pending_loss = (ts, scales, float(len(ts)), out)
            return out
        cell = StepScope.early_loss if isinstance(StepScope.early_loss, K.HostCell) else None
        # (round 4, measured and removed: this 8 us single-block launch on a stream of its own, so that it does not stand
        # between the exits' forward and their backward -- step 1.631-1.657 against 1.628-1.635 ms: a third branch in the
        # captured graph costs more than the launch, profiles/r04_ab_loss_side.txt)
        if len(ts) <= 8:
            return K.loss_from_partials(ts, scales, float(len(ts)), host_cell=cell)
        # more terms than one launch takes: sums of groups of 8 first, then their mean
        groups = [K.loss_from_partials(ts[i:i + 8], scales[i:i + 8], 1.0) for i in range(0, len(ts), 8)]
        return K.loss_from_partials(groups, [1.0] * len(groups), float(len(ts)), host_cell=cell)

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.meta)
        gm = None
        grads = []
        for (scale, prescaled), shape in zip(ctx.meta, ctx.shapes):
            if prescaled:
                grads.append(g.expand(shape))
            else:
                gm = g / n if gm is None else gm
                grads.append(gm if scale == 1.0 else gm * scale)
        return (None,) + tuple(grads)


def mean_of_terms(terms):
    """terms: LossTerm or 0-d tensors -> their mean as a 0-d tensor (one launch)."""
    terms = [t if isinstance(t, LossTerm) else LossTerm(t) for t in terms]
    return MeanTermsFn.apply([(t.scale, t.prescaled) for t in terms], *[t.tensor for t in terms])


class MergeFn(torch.autograd.Function):
    """LarvaTail's torch.cat(features, 1) + merge_conv (models/LarvaNetV2.py:328-330) without
    materialising the concatenation: the conv kernel walks the feature tensors as K chunks."""

    @staticmethod
    def forward(ctx, pc, weight, bias, *feats):
        packs = pc.get()
        cout = int(weight.shape[0])
        out = K.conv3x3(list(feats), packs[0][0], cout, bias=bias.detach(), logical_w=_lw())
        ctx.save_for_backward(*feats)
        ctx.pc = pc
        ctx.wshape = tuple(weight.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        feats = ctx.saved_tensors
        packs = ctx.pc.get()
        dy = dy.contiguous()
        cout = ctx.wshape[0]
        c = int(feats[0].shape[1])
        dfeats = [K.conv3x3(dy, packs[1 + i][1], c) for i in range(len(feats))]
        tw, tb = _targets(ctx.pc)
        dw = tw if tw is not None else torch.empty(ctx.wshape, device=dy.device, dtype=torch.float32)
        jobs = [(dy, f, ctx.wshape, i * c, c, dw, tb if i == 0 else None) for i, f in enumerate(feats)]
        res = _wgrad(jobs, cout, c, inplace=tw is not None)
        db = res[0][1]
        if tw is not None:
            return (None, None, None) + tuple(dfeats)
        return (None, dw, db) + tuple(dfeats)


class L1LossFn(torch.autograd.Function):
    """nn.L1Loss() (models/LarvaNet.py:85,108)."""

    @staticmethod
    def forward(ctx, out, truth):
        out = out.contiguous()
        truth = truth.contiguous()
        ctx.save_for_backward(out, truth)
        return K.l1_fwd(out, truth)

    @staticmethod
    def backward(ctx, g):
        out, truth = ctx.saved_tensors
        return K.l1_bwd(out, truth, g.contiguous()), None
