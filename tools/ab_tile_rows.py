#!/usr/bin/env python3
"""Round 4: 3 x 48 against 4 x 48 tiles on a full 339 x 510 image (pitch 512): per-layer time of a captured chain of 34
conv + ReLU launches (the inference forward's body) and the whole fwd_runtime of V1 / V2."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for C in (48, 32):
    x = (torch.randn(1, C, 339, 512, generator=g) * 20).to(dev)
    x[..., 510:] = 0
    bufs = [x, torch.empty_like(x)]
    fwd, _ = K.pack_weights((torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev))
    b = torch.zeros(C, device=dev)
    for rows in (3, 4, 0):
        def chain():
            for i in range(34):
                K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], logical_w=510, tile_rows=rows)
        chain()
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            chain()
        for _ in range(3):
            gph.replay()
        torch.cuda.synchronize()
        runs = []
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                gph.replay()
            e.record()
            torch.cuda.synchronize()
            runs.append(s.elapsed_time(e) / 10 / 34 * 1e3)
        flop = 2 * 9 * C * C * 339 * 510
        us = sorted(runs)[1]
        print("%d channels, 1 x %d x 339 x 510, tile_rows=%d: %.2f us per conv+ReLU layer = %.1f TFLOP/s = %.3f of the fp32 matrix peak"
              % (C, C, rows, us, flop / us / 1e6, flop / us / 1e6 / 157.3))
