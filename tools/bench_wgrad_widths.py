#!/usr/bin/env python3
"""bench.py's weight-gradient block (one launch over 40 layers + reduction, captured and replayed) at the given channel
counts, and the partial-image launch alone.  LARVA_WGRAD_PIPE=0 in the environment gives the register-staged kernel of
every shape, LARVA_HIP_LIB a variant build, for same-box A/Bs.
usage: bench_wgrad_widths.py [C ...]   (default 32 48 64)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)


def flat_alone(c, jobs=40, iters=10):
    g = torch.Generator().manual_seed(6)
    dy = (torch.randn(16, c, 48, 48, generator=g) * 1e-3).to(dev)
    xs = (torch.randn(16, c, 48, 48, generator=g) * 20).to(dev)
    js = [{"dy": dy + 0, "x": xs + 0} for _ in range(jobs)]
    if K.conv3x3_wgrad_partial_flat(js, c, c, 256) is None:
        return None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = K.conv3x3_wgrad_partial_flat(js, c, c, 256)
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


for c in [int(a) for a in sys.argv[1:]] or [32, 48, 64]:
    b = bench.wgrad_block(dev, c)
    alone = flat_alone(c)
    print("C=%d  %.2f us per layer  %.1f TFLOP/s = %.3f of peak;  partial-image launch alone %s us per layer  (%s)"
          % (c, b["ms_per_layer"] * 1e3, b["achieved"], b["frac"], "%.2f" % alone if alone else "-", b["kernel"][:40]), flush=True)
