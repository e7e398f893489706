#!/usr/bin/env python3
"""The step prologue launch (weight packing + head input padding + bicubic base) alone, back to back, and its
parts as launches of their own (MI355X: 10.6 us fused against 7.4 + 6.5 us of the parts; 14.3 us while every
packed float was gathered by a thread of its own, 8640 workgroups that did not fit the chip at once)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K
from larvanet_amd.autograd import StepScope

dev = torch.device("cuda", 0)
m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
torch.manual_seed(0)
m.prepare(is_training=True, scales=[4])
x = (torch.rand(16, 3, 48, 48) * 255).to(dev)
jobs = []
for pc in m.model.packed_convs():
    jobs += pc.jobs()
x16 = torch.zeros((16, 16, 48, 48), device=dev)
base = torch.empty((16, 3, 192, 192), device=dev)
flush = torch.empty(64 << 20, device=dev)


def graph_of(fn, reps):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return g


def timed(fn, reps=20):
    """us per fn() inside a replayed graph (the Python wrapper's own time would hide a 10 us kernel)."""
    g = graph_of(fn, reps)
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (10 * reps)


fill = timed(lambda: flush.fill_(1.0))
full = timed(lambda: K.step_prologue(jobs, x, x16, base))
cold = timed(lambda: (flush.fill_(1.0), K.step_prologue(jobs, x, x16, base))) - fill
print("prologue launch %.2f us back to back, %.2f after a 256 MB fill" % (full, cold), flush=True)
print("pack_weights_batch alone %.2f us, bicubic4 alone %.2f us" % (timed(lambda: K.pack_weights_batch(jobs)), timed(lambda: K.bicubic4(x))))
