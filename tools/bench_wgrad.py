#!/usr/bin/env python3
"""Weight-gradient kernel alone: us per layer at a given batching, with the operands either hot
(the same tensors every iteration: they sit in the 256 MB Infinity Cache) or cold (a 512 MB
buffer is rewritten between iterations, as a training step's other kernels do).
usage: bench_wgrad.py [jobs] [splits] [iters] [cold|hot] [C]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 256 // jobs
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cold = (sys.argv[4] if len(sys.argv) > 4 else "cold") == "cold"
C = int(sys.argv[5]) if len(sys.argv) > 5 else 48

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
dys = [(torch.randn(16, C, 48, 48, generator=g) * 1e-3).to(dev) for _ in range(jobs)]
xs = [(torch.randn(16, C, 48, 48, generator=g) * 20).to(dev) for _ in range(jobs)]
js = [{"dy": dys[i], "x": xs[i], "dw": torch.empty(C, C, 3, 3, device=dev), "db": torch.empty(C, device=dev)}
      for i in range(jobs)]
parts = K.conv3x3_wgrad(js, C, C, splits)
for j, p in zip(js, parts):
    j["partial"] = p
scratch = torch.empty(128 * 1024 * 1024, device=dev)


def loop(with_wgrad):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for it in range(iters):
        if cold:
            scratch.fill_(float(it))
        if with_wgrad:
            K.conv3x3_wgrad(js, C, C, splits)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


loop(True)
med = min(loop(True) for _ in range(3)) - (min(loop(False) for _ in range(3)) if cold else 0.0)
flop = 2 * 9 * C * C * 16 * 48 * 48 * jobs
print("wgrad+reduce  C=%d jobs=%d splits=%d %s: %.1f us per launch pair = %.2f us/layer = %.1f TFLOP/s (%.0f %% of 157.3)"
      % (C, jobs, splits, "cold" if cold else "hot", med, med / jobs, flop / med / 1e6, flop / med / 1e6 / 1.573))
