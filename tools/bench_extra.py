#!/usr/bin/env python3
"""Extra measurements quoted in DESIGN.md: the conv / wgrad kernels at 32, 48 and 64 channels
(SURVEY 8a note N1) and full-image inference (validate.py path) at a DIV2K-val-like size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def conv_us(C, iters=40):
    x = (torch.randn(16, C, 48, 48, generator=g) * 20).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(C, device=dev)
    fwd, _ = K.pack_weights(w)
    out = torch.empty_like(x)
    K.conv3x3(x, fwd, C, bias=b, relu=True, out=out)
    import diag_lib   # (tools/diag_lib.py: kernel-attached timing lives in the measurement library)
    mean, _ = diag_lib.conv3x3_relu_timed(x, fwd, C, b, out, iters)
    return mean * 1e3


def wgrad_us(C, jobs=8, iters=10):
    dys = [(torch.randn(16, C, 48, 48, generator=g) * 1e-3).to(dev) for _ in range(jobs)]
    xs = [(torch.randn(16, C, 48, 48, generator=g) * 20).to(dev) for _ in range(jobs)]
    js = [{"dy": dys[i], "x": xs[i], "dw": torch.empty(C, C, 3, 3, device=dev), "db": torch.empty(C, device=dev)}
          for i in range(jobs)]
    splits = 256 // jobs
    parts = K.conv3x3_wgrad(js, C, C, splits)
    for j, p in zip(js, parts):
        j["partial"] = p
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        K.conv3x3_wgrad(js, C, C, splits)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3 / jobs


print("channels  conv+relu us  TFLOP/s  frac | wgrad+reduce us per layer (8-layer batch)  TFLOP/s")
for C in (32, 48, 64):
    flop = 2 * 9 * C * C * 16 * 48 * 48
    cu = conv_us(C)
    wu = wgrad_us(C)
    print("%5d %12.1f %9.1f %6.3f | %12.1f %9.1f" % (C, cu, flop / cu / 1e6, flop / cu / 1e6 / 157.3, wu, flop / wu / 1e6))

import importlib
for name, h, w in (("LarvaNet", 339, 510), ("LarvaNetV2", 339, 510), ("LarvaNet", 340, 512)):
    m = importlib.import_module("larvanet_amd.models." + name).create_model()
    m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
    torch.manual_seed(0)
    m.prepare(is_training=False, scales=[4])
    x = (torch.rand(1, 3, h, w, generator=g) * 255).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m.model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.model(x)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("%s full image %dx%d -> %dx%d: %.2f ms (%.1f HR Mpix/s)" % (name, h, w, 4 * h, 4 * w, ms, 16 * h * w / ms / 1e3))
