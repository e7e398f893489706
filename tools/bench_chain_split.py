#!/usr/bin/env python3
"""How should the 16 images of a training batch be cut into layer chains?  (Round 5)  Chains of
strip-tile launches over 16 images in all, 40 links, captured: one chain of 16, 8 + 8 (the product's), 6 + 5 + 5,
4 x 4; per width.  us per full-batch layer and the fraction of the fp32 matrix peak.

  python tools/bench_chain_split.py [widths ...]        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
LAYERS = 40
streams = [torch.cuda.Stream() for _ in range(8)]
SPLITS = {"16": [(0, 16)], "8+8": [(0, 8), (8, 16)], "6+5+5": [(0, 6), (6, 11), (11, 16)], "4x4": [(0, 4), (4, 8), (8, 12), (12, 16)],
          "8x2": [(2 * k, 2 * k + 2) for k in range(8)]}


def setup(c, n):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
    fwd, _ = K.pack_weights(w)
    bufs = [(torch.randn(n, c, 48, 48, generator=g) * 20).to(dev), torch.empty(n, c, 48, 48, device=dev)]
    return fwd, torch.zeros(c, device=dev), bufs


def chains(c, fwd, b, bufs, parts, whole=False):
    cur = torch.cuda.current_stream()
    if whole:   # the whole-batch launch of 3 x 48 tiles (256 workgroups), one chain
        for i in range(LAYERS):
            K.conv3x3(bufs[i & 1], fwd, c, bias=b, relu=True, out=bufs[(i + 1) & 1])
        return
    for k, rng in enumerate(parts):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for i in range(LAYERS):
                K.conv3x3(bufs[i & 1], fwd, c, bias=b, relu=True, out=bufs[(i + 1) & 1], images=rng, strips=2 if k & 1 else True,
                          plain_stores=True)
    for k in range(len(parts)):
        cur.wait_stream(streams[k])


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        fn()
    for _ in range(5):
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
        best.append(s.elapsed_time(e) / reps * 1e3 / LAYERS)
    return sorted(best)[2]


for c in [int(v) for v in sys.argv[1:]] or (32, 48, 64):
    K.strip_tile_table(48, 48, dev, 0)
    K.strip_tile_table(48, 48, dev, 1)
    fwd, b, bufs = setup(c, 16)
    flop = 2 * 9 * c * c * 16 * 48 * 48
    row = ["3x48 tiles, one launch %.2f us (%.3f)" % ((t := timed(lambda: chains(c, fwd, b, bufs, None, True))), flop / t / 157.3e6)]
    for name, parts in SPLITS.items():
        t = timed(lambda: chains(c, fwd, b, bufs, parts))
        row.append("%s %.2f us (%.3f)" % (name, t, flop / t / 157.3e6))
    print("%d channels, 16 x %d x 48 x 48, us per full-batch layer (fraction of 157.3 TFLOP/s): " % (c, c) + " | ".join(row))
