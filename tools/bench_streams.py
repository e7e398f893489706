#!/usr/bin/env python3
"""Do two independent conv chains on two HIP streams overlap on the chip?  (The LDS-DMA conv
kernel is sized so that two workgroups share a CU.)  Prints wall time of: one chain alone, two
chains back to back on one stream, two chains on two streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
w = (torch.randn(48, 48, 3, 3, generator=g) * 0.02).to(dev)
b = torch.zeros(48, device=dev)
fwd, _ = K.pack_weights(w)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
xs = [(torch.randn(N, 48, 48, 48, generator=g) * 20).to(dev) for _ in range(2)]
ys = [torch.empty_like(x) for x in xs]
LAYERS = 20


def chain(i):
    a, c = xs[i], ys[i]
    for _ in range(LAYERS):
        K.conv3x3(a, fwd, 48, bias=b, relu=True, out=c)
        a, c = c, a


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def graphed(fn):
    gph = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(gph):
        fn()
    return gph.replay


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(0)
    with torch.cuda.stream(s2):
        chain(1)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


one = timed(graphed(lambda: chain(0)))
serial = timed(graphed(lambda: (chain(0), chain(1))))
par = timed(graphed(two_streams))
print("batch %d: one chain of %d convs: %.1f us (%.2f us/conv); two chains serial: %.1f us; two streams: %.1f us "
      "(%.2f us per conv-unit)" % (N, LAYERS, one, one / LAYERS, serial, par, par / (2 * LAYERS)))
