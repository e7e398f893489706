#!/usr/bin/env python3
"""One batched launch of k independent convs vs k launches (16x48x48x48, conv+ReLU), us."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
C = 48


def job():
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    fwd, _ = K.pack_weights(w)
    return {"srcs": (torch.randn(16, C, 48, 48, generator=g) * 20).to(dev), "wpk": fwd, "bias": torch.zeros(C, device=dev)}


def timed(fn, iters=50):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / iters)
    return best


for k in (2, 3, 4):
    jobs = [job() for _ in range(k)]
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            for j in jobs:
                K.conv3x3(j["srcs"], j["wpk"], C, bias=j["bias"], relu=True)
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb):
        for _ in range(10):
            K.conv3x3_batch(jobs, C, relu=True)
    print("%d jobs: separate launches %.1f us, one batched launch %.1f us" % (k, timed(gr.replay, 20) / 10, timed(gb.replay, 20) / 10))
