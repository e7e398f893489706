#!/usr/bin/env python3
"""One DIV2K-val-sized LR image (1 x 48 x 339 x 510, rows padded to 512): 3 x 48 tiles (1243 workgroups per
launch) against strip tiles (5 x 16 / 4 x 16: 2400 workgroups), chain of 36 conv+ReLU launches in a graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
H, W, P, C, LAYERS = 339, 510, 512, 48, 36
w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).to(dev)
b = torch.zeros(C, device=dev)
fwd, _ = K.pack_weights(w)
x = torch.zeros(1, C, H, P, device=dev)
x[..., :W] = (torch.randn(1, C, H, W, generator=g) * 20).to(dev)
bufs = [x, torch.zeros_like(x)]
print("strip table:", K.strip_tile_table(H, P, dev)[1], "tiles per image")


def chain(strips, plain):
    for i in range(LAYERS):
        K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], logical_w=W, strips=strips, plain_stores=plain)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        fn()
    for _ in range(3):
        gph.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            gph.replay()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / 5 * 1e3 / LAYERS)
    return sorted(best)[1]


flop = 2 * 9 * C * C * H * W
for name, s, pl in (("3 x 48 tiles", False, False), ("strip tiles, non-temporal stores", True, False), ("strip tiles, plain stores", True, True)):
    t = timed(lambda: chain(s, pl))
    print("%-36s %.1f us per layer = %.1f TFLOP/s" % (name, t, flop / t / 1e6))
