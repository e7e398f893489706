#!/usr/bin/env python3
"""How does a hipGraph replay schedule two parallel branches?  A step-like graph (a few whole-batch
kernels, then two half-batch chains of CHAIN strip-tile convs, then whole-batch kernels again) replayed
(a) as ONE captured graph with a fork / join, (b) as per-chain graphs launched on two explicit streams,
(c) as (b) cut into segments of SEG layers launched alternately.  Wall time per replay with the host
kept from running ahead (a sync between replays) and back to back; host time of the launch call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
N, C, CHAIN = 16, 48, int(os.environ.get("CHAIN", "33"))
w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).to(dev)
b = torch.zeros(C, device=dev)
fwd, _ = K.pack_weights(w)
bufs = [(torch.randn(N, C, 48, 48, generator=g) * 20).to(dev), torch.empty(N, C, 48, 48, device=dev)]
K.strip_tile_table(48, 48, dev, 0)
K.strip_tile_table(48, 48, dev, 1)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
PARTS = ((0, N // 2), (N // 2, N))


def whole(n=2):
    for i in range(n):
        K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1])


def chain(k, lo, hi):
    for i in range(lo, hi):
        K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], images=PARTS[k], strips=2 if k else True)


def one_graph_body():
    cur = torch.cuda.current_stream()
    whole()
    sA.wait_stream(cur)
    sB.wait_stream(cur)
    for i in range(CHAIN):      # issue order A0 B0 A1 B1 ...
        for k, st in enumerate((sA, sB)):
            with torch.cuda.stream(st):
                chain(k, i, i + 1)
    cur.wait_stream(sA)
    cur.wait_stream(sB)
    whole()


def capture(fn, stream=None):
    fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    if stream is None:
        with torch.cuda.graph(gph):
            fn()
    else:
        with torch.cuda.graph(gph, stream=stream):
            fn()
    return gph


def measure(name, launch, reps=30):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    # (1) a sync between replays: the host cannot run ahead
    t_sync, t_host = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launch()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t_sync.append(time.perf_counter() - t0)
        t_host.append(t1 - t0)
    # (2) back to back
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / reps
    print("%-58s sync'd %7.1f us   back to back %7.1f us   host launch call %6.1f us"
          % (name, sorted(t_sync)[reps // 2] * 1e6, b2b * 1e6, sorted(t_host)[reps // 2] * 1e6))


ONLY = os.environ.get("ONLY", "")
if ONLY:
    g_pre, g_post = capture(whole), capture(whole)
    if ONLY == "one":
        one = capture(one_graph_body)
        measure("two chains, ONE graph with fork/join", one.replay)
    else:
        seg = int(ONLY)
        cuts = list(range(0, CHAIN, seg)) + [CHAIN]
        segs = [[capture(lambda k=k, lo=lo, hi=hi: chain(k, lo, hi), stream=(sA, sB)[k]) for k in range(2)]
                for lo, hi in zip(cuts[:-1], cuts[1:])]

        def launch(segs=segs):
            cur = torch.cuda.current_stream()
            g_pre.replay()
            sA.wait_stream(cur)
            sB.wait_stream(cur)
            for ga, gb in segs:
                with torch.cuda.stream(sA):
                    ga.replay()
                with torch.cuda.stream(sB):
                    gb.replay()
            cur.wait_stream(sA)
            cur.wait_stream(sB)
            g_post.replay()

        measure("two chains, per-chain graphs in segments of %d layers" % seg, launch)
    sys.exit(0)

single = capture(lambda: (whole(), [K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1]) for i in range(CHAIN)], whole()))
measure("one chain of whole-batch launches (one graph)", single.replay)

one = capture(one_graph_body)
measure("two chains, ONE graph with fork/join", one.replay)

g_pre, g_post = capture(whole), capture(whole)
for seg in (CHAIN, 8, 4):
    cuts = list(range(0, CHAIN, seg)) + [CHAIN]
    segs = [[capture(lambda k=k, lo=lo, hi=hi: chain(k, lo, hi), stream=(sA, sB)[k]) for k in range(2)]
            for lo, hi in zip(cuts[:-1], cuts[1:])]

    def launch(segs=segs):
        cur = torch.cuda.current_stream()
        g_pre.replay()
        sA.wait_stream(cur)
        sB.wait_stream(cur)
        for ga, gb in segs:
            with torch.cuda.stream(sA):
                ga.replay()
            with torch.cuda.stream(sB):
                gb.replay()
        cur.wait_stream(sA)
        cur.wait_stream(sB)
        g_post.replay()

    measure("two chains, per-chain graphs in segments of %d layers" % seg, launch)


def eager():
    cur = torch.cuda.current_stream()
    whole()
    sA.wait_stream(cur)
    sB.wait_stream(cur)
    for i in range(CHAIN):
        for k, st in enumerate((sA, sB)):
            with torch.cuda.stream(st):
                chain(k, i, i + 1)
    cur.wait_stream(sA)
    cur.wait_stream(sB)
    whole()


measure("two chains, eager launches on two streams (no graph)", eager)
