#!/usr/bin/env python3
"""What a (48, 16) head tile costs beside a (48, 48) tile with ONE workgroup per CU (the flat grid's tail): 8 layers x 32
workgroups of each shape, partial-image launch only, captured and replayed.   LARVA_HIP_LIB selects a variant build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)


def time_shape(cout, cin, jobs=8, splits=32, iters=10):
    g = torch.Generator().manual_seed(6)
    dy = (torch.randn(16, cout, 48, 48, generator=g) * 1e-3).to(dev)
    xs = (torch.randn(16, cin, 48, 48, generator=g) * 20).to(dev)
    js = [{"dy": dy + 0, "x": xs + 0} for _ in range(jobs)]
    K.conv3x3_wgrad_partial(js, cout, cin, splits)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = K.conv3x3_wgrad_partial(js, cout, cin, splits)
    for _ in range(3):
        graph.replay()
    runs = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            graph.replay()
        e.record()
        torch.cuda.synchronize()
        runs.append(s.elapsed_time(e) / iters)
    del keep
    return sorted(runs)[1] * 1e3 / jobs


full = time_shape(48, 48)
for cout, cin in ((48, 16), (48, 32), (32, 16), (64, 16)):
    t = time_shape(cout, cin)
    print("(%d, %d): %.2f us per tile = %.2f of a (48, 48) tile (%.2f us); MFMA count ratio %.2f"
          % (cout, cin, t, t / full, full, cout * cin / (48.0 * 48.0)), flush=True)
