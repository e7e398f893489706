#!/usr/bin/env python3
"""The body of a full-image forward (M4B4: 32 body convs + the last leg's first conv on 1 x 48 x 339 x 510, pitch 512) as
ONE layer-pipeline launch (tools/diag_lib.ConvPipeline: the measurement library) against one persistent launch per layer, both as captured graphs.
  python tools/bench_pipeline.py [H W [spin_limit]]        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from larvanet_amd import kernels as K
import diag_lib

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (339, 510)
dev = torch.device("cuda", 0)
P = (W + 3) // 4 * 4
g = torch.Generator().manual_seed(0)
x = torch.zeros(1, 48, H, P, device=dev)
x[..., :W] = torch.randn(1, 48, H, W, generator=g).to(dev)
specs, cur = [], -1
for m in range(4):
    m_in = cur
    for j in range(4):
        b_in = cur
        specs.append(([cur], True, None, None))
        specs.append(([len(specs) - 1], False, b_in, m_in if j == 3 else None))
        cur = len(specs) - 1
specs.append(([cur], True, None, None))
weights = []
for srcs, relu, r0, r1 in specs:
    fwd, _ = K.pack_weights((torch.randn(48, 48, 3, 3, generator=g) * (1.0 / 432) ** 0.5).to(dev))
    weights.append((fwd, (torch.randn(48, generator=g) * 0.1).to(dev)))
outs = [torch.empty(1, 48, H, P, device=dev) for _ in specs]
refs = [torch.empty(1, 48, H, P, device=dev) for _ in specs]
u = lambda i: x if i == -1 else outs[i]
t = lambda i: x if i == -1 else refs[i]
layers = [dict(srcs=[u(i) for i in srcs], wpk=fwd, bias=b, relu=relu, res0=None if r0 is None else u(r0),
               res1=None if r1 is None else u(r1), out=outs[k], dep=max(srcs))
          for k, ((srcs, relu, r0, r1), (fwd, b)) in enumerate(zip(specs, weights))]
pipe = diag_lib.ConvPipeline(layers, logical_w=W)


def per_layer():
    for k, ((srcs, relu, r0, r1), (fwd, b)) in enumerate(zip(specs, weights)):
        K.conv3x3([t(i) for i in srcs], fwd, 48, bias=b, relu=relu, res0=None if r0 is None else t(r0),
                  res1=None if r1 is None else t(r1), logical_w=W, out=refs[k])


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        fn()
    for _ in range(3):
        gph.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            gph.replay()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / reps)
    return sorted(best)[2]


flop = 2 * 9 * 48 * 48 * H * W * len(specs)
tp = timed(pipe.run)
pipe.check()
tl = timed(per_layer)
same = all(torch.equal(a, b) for a, b in zip(outs, refs))
print("%d layers on 1 x 48 x %d x %d: pipeline %.3f ms = %.1f us per layer (%.3f of the fp32 matrix peak) | one persistent launch per layer %.3f ms = %.1f us "
      "per layer (%.3f) | outputs identical: %s" % (len(specs), H, W, tp, tp * 1e3 / len(specs), flop / tp / 157.3e9, tl, tl * 1e3 / len(specs),
                                                     flop / tl / 157.3e9, same))
