#!/usr/bin/env python3
"""Medium-size shape fuzz (round 5): one training step (loss + every gradient) and upscale() at shapes between the toy
fixtures and the headline batch -- strips / two chains, whole-batch launches, persistent tiles, odd widths -- V1 and V2,
eager launches against the captured graph ON THE SAME WEIGHTS (same forward bits, so the same ReLU masks and L1 signs:
what is left is the summation order of the weight gradients, ~1e-6), and both against oracle/larva_torch.py in fp32 and
fp64 (where an fp32 implementation is ~1e-3 from fp64 so is torch's own CPU fp32 run: a ReLU mask or an L1 sign that flips).
  python tools/fuzz_medium.py [num_filters=48]       (GPU box; needs oracle/, so it is a tool beside the tests, not product code)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import larva_torch as T

dev = torch.device("cuda", 0)
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 48


def model(name, argv, seed):
    m = importlib.import_module("larvanet_amd.models." + name).create_model()
    m.parse_args(argv)
    torch.manual_seed(seed)
    m.prepare(is_training=True, scales=[4])
    return m


rng = np.random.RandomState(99)
cases = [(2, 52, 52), (4, 52, 52), (16, 24, 24), (3, 100, 100), (1, 200, 200), (2, 150, 150), (5, 64, 64), (8, 96, 96), (2, 48, 130), (3, 61, 47),
         (7, 33, 58), (16, 48, 48)]
if os.environ.get("FUZZ_CASES"):   # "n,h,w;n,h,w;..."
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["FUZZ_CASES"].split(";")]
bad = 0
for ci, (n, h, w) in enumerate(cases):
    for v2 in (False, True):
        name = "LarvaNetV2" if v2 else "LarvaNet"
        blocks = [1, 2] if ci % 2 else [2]
        flags = ["--num_modules=%d" % len(blocks), "--num_blocks=%s" % ",".join(map(str, blocks))] + (["--num_filters=%d" % NF] if NF != 48 else [])
        m = model(name, flags, 10 + ci)
        sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
        x = torch.from_numpy(rng.randint(0, 256, size=(n, 3, h, w)).astype(np.float32))
        t = torch.from_numpy(rng.randint(0, 256, size=(n, 3, 4 * h, 4 * w)).astype(np.float32))
        grads, losses = {}, {}
        for graph in (False, True):
            m.use_hip_graph = graph
            for _ in range(3 if graph else 1):   # (the third call replays the captured step)
                loss, _ = m._forward_backward(x.to(dev), t.to(dev))
            torch.cuda.synchronize()
            grads[graph] = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.model.named_parameters()}
            losses[graph] = float(loss.detach())
        sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        r32 = T.multi_exit_loss(sd32, x, t, blocks, v2=v2)
        r32.backward()
        sd64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
        r64 = T.multi_exit_loss(sd64, x.double(), t.double(), blocks, v2=v2)
        r64.backward()

        def rel(a, b):
            return max(float(np.abs(a[k].astype(np.float64) - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)) for k in a)

        g32 = {k: sd32[k].grad.numpy() for k in sd32}
        g64 = {k: sd64[k].grad.numpy() for k in sd64}
        e_self = rel(grads[True], {k: v.astype(np.float64) for k, v in grads[False].items()})
        e_gpu, e_cpu = max(rel(grads[False], g64), rel(grads[True], g64)), rel(g32, g64)
        e_loss = max(abs(losses[g] - float(r64.detach())) / abs(float(r64.detach())) for g in (False, True))
        with torch.no_grad():
            got = m.upscale([x[i].numpy() for i in range(n)], 4)
            ref = (T.forward_v2(sd, x, blocks) if v2 else T.forward(sd, x, blocks)).numpy()
        e_fwd = float(np.abs(got - ref).max())
        # eager == graph to summation-order noise.  Against fp64 only a gross bar: a ReLU mask or an L1 sign that flips at a
        # near-zero value moves a (cancelling) bias-gradient sum by 1e-3 of its maximum, and it happens to either fp32
        # implementation independently -- the torch CPU fp32 column is there to read the gpu column against
        ok = e_self <= 2e-5 and e_loss <= 2e-5 and e_fwd <= 2e-3 and e_gpu <= 2e-2
        bad += not ok
        print("%-10s %2d x 3 x %3d x %3d blocks %s: graph vs eager %.1e | vs fp64: gpu %.1e, torch cpu fp32 %.1e | loss %.1e | upscale %.1e  %s"
              % (name, n, h, w, blocks, e_self, e_gpu, e_cpu, e_loss, e_fwd, "ok" if ok else "BAD"), flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
