#!/usr/bin/env python3
"""Would a THIRD co-resident workgroup per CU help the layer chains?  (Round 4.)  At 48 channels a strip workgroup holds
60 KB of LDS (3 stages x (6 KB halo tile + 13.5 KB weights)): two per CU.  At 32 channels it holds 45 KB: THREE fit.  So
the question can be asked of the product kernel without rebuilding it: chains of 8-image strip launches (256 workgroups
each), two chains against three, per 8-image launch.  If three chains at 32 channels finish a layer of 24 images in
about the time two chains finish 16, a chain is latency-bound and a third resident workgroup is worth a kernel whose
weights do not live in the LDS ring; if the time grows by 3/2, the CU is already busy.

  python tools/bench_chain_count.py        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
LAYERS = 40
streams = [torch.cuda.Stream() for _ in range(4)]


def setup(c, n):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
    fwd, _ = K.pack_weights(w)
    bufs = [(torch.randn(n, c, 48, 48, generator=g) * 20).to(dev), torch.empty(n, c, 48, 48, device=dev)]
    return fwd, torch.zeros(c, device=dev), bufs


def chains(c, fwd, b, bufs, parts):
    cur = torch.cuda.current_stream()
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
    for _ in range(3):
        gph.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            gph.replay()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / reps * 1e3 / LAYERS)
    return sorted(best)[1]


for c in (32, 48, 64):
    K.strip_tile_table(48, 48, dev, 0)
    K.strip_tile_table(48, 48, dev, 1)
    fwd, b, bufs = setup(c, 32)
    res = {}
    for nch in (1, 2, 3, 4):
        parts = [(8 * k, 8 * k + 8) for k in range(nch)]
        res[nch] = timed(lambda: chains(c, fwd, b, bufs, parts))
    print("%d channels, chains of 8-image strip launches (256 workgroups each), us per layer of ALL chains / per 8-image launch:" % c)
    for nch in (1, 2, 3, 4):
        print("   %d chain%s  %6.2f us per layer   %6.2f us per 8 images   (x%.2f of two chains' cost per image)"
              % (nch, "s" if nch > 1 else " ", res[nch], res[nch] / nch, (res[nch] / nch) / (res[2] / 2)))
