#!/usr/bin/env python3
"""One dependent chain of full-batch conv launches (3 x 48 tiles, 256 workgroups each) against the
same layers run as TWO half-batch chains of strip-tile launches (5 x 16 / 4 x 16 tiles, 256
workgroups each) on two streams -- both captured in a hipGraph, per full-batch layer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
N, C, LAYERS = 16, 48, 40
w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).to(dev)
b = torch.zeros(C, device=dev)
fwd, _ = K.pack_weights(w)
bufs = [(torch.randn(N, C, 48, 48, generator=g) * 20).to(dev), torch.empty(N, C, 48, 48, device=dev)]
K.strip_tile_table(48, 48, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def single(strips=False):
    for i in range(LAYERS):
        K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], strips=strips)


def dual(strips=True, split=N // 2):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    for st, rng in ((s1, (0, split)), (s2, (split, N))):
        with torch.cuda.stream(st):
            for i in range(LAYERS):
                K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], images=rng, strips=strips)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


streams = [torch.cuda.Stream() for _ in range(4)]
scratch = [torch.empty_like(bufs[0]) for _ in range(2)]


def multi(parts, stagger=0, alt=False):
    """len(parts) chains over image ranges `parts`; chain k first runs `stagger * k` extra launches on
    a scratch buffer (a phase offset between the chains)."""
    cur = torch.cuda.current_stream()
    for k, rng in enumerate(parts):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for _ in range(stagger * k):
                K.conv3x3(scratch[0], fwd, C, bias=b, relu=True, out=scratch[1], images=(0, 4), strips=True)
            for i in range(LAYERS):
                K.conv3x3(bufs[i & 1], fwd, C, bias=b, relu=True, out=bufs[(i + 1) & 1], images=rng,
                          strips=2 if (alt and k & 1) else True)
    for k in range(len(parts)):
        cur.wait_stream(streams[k])


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        fn()
    return gph.replay


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / reps * 1e3 / LAYERS)
    return sorted(best)[1]


print("us per full-batch 48->48 conv+ReLU layer (16x48x48x48), median of 3 x 20 graph replays of %d layers:" % LAYERS)
print("  one chain, 3x48 tiles (256 WG / launch)          %.2f" % timed(graphed(single)))
print("  one chain, strip tiles (512 WG / launch)         %.2f" % timed(graphed(lambda: single(True))))
print("  two half-batch chains, strip tiles, two streams  %.2f" % timed(graphed(dual)))
print("  two half-batch chains, 3x48 tiles, two streams   %.2f" % timed(graphed(lambda: dual(False))))
print("  two chains, chain 1 delayed by one 4-image launch  %.2f" % timed(graphed(lambda: multi([(0, 8), (8, 16)], 1))))
print("  two chains, chain 1 delayed by two 4-image launches %.2f" % timed(graphed(lambda: multi([(0, 8), (8, 16)], 2))))
print("  four quarter-batch chains                           %.2f" % timed(graphed(lambda: multi([(0, 4), (4, 8), (8, 12), (12, 16)]))))
print("  four quarter-batch chains, staggered                %.2f" % timed(graphed(lambda: multi([(0, 4), (4, 8), (8, 12), (12, 16)], 1))))
print("  two chains 9 + 7 images                             %.2f" % timed(graphed(lambda: multi([(0, 9), (9, 16)]))))
print("  two chains, complementary tile tables               %.2f" % timed(graphed(lambda: multi([(0, 8), (8, 16)], 0, True))))


# the same two chains the way a training step sees them: every layer writes a tensor of its own (kept for
# backward) and reads weights of its own, with the step's epilogue mix (ReLU / +res0 / +res0+res1)
acts = [bufs[0]] + [torch.empty_like(bufs[0]) for _ in range(LAYERS)]
wpks = [K.pack_weights((torch.randn(C, C, 3, 3, generator=g) * 0.02).to(dev))[0] for _ in range(LAYERS)]


def step_like(parts, epi_mix, distinct, own_weights=True):
    cur = torch.cuda.current_stream()
    for k, rng in enumerate(parts):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for i in range(LAYERS):
                src = acts[i] if distinct else bufs[i & 1]
                dst = acts[i + 1] if distinct else bufs[(i + 1) & 1]
                kw = {"relu": True}
                if epi_mix and i % 2 == 1:
                    prev = acts[i - 1] if distinct else bufs[(i - 1) & 1]
                    kw = {"res0": prev} if i % 8 != 7 else {"res0": prev, "res1": acts[max(i - 7, 0)] if distinct else bufs[0]}
                K.conv3x3(src, wpks[i] if (distinct and own_weights) else fwd, C, bias=b, out=dst, images=rng, strips=2 if k else True, **kw)
    for k in range(len(parts)):
        cur.wait_stream(streams[k])


HALVES = [(0, 8), (8, 16)]
print("  two chains, ReLU only, ping-pong buffers               %.2f" % timed(graphed(lambda: step_like(HALVES, False, False))))
print("  two chains, ReLU only, a tensor + weights per layer    %.2f" % timed(graphed(lambda: step_like(HALVES, False, True))))
print("  two chains, step's epilogue mix, ping-pong             %.2f" % timed(graphed(lambda: step_like(HALVES, True, False))))
print("  two chains, step's epilogue mix, tensor per layer      %.2f" % timed(graphed(lambda: step_like(HALVES, True, True))))
print("  two chains, ReLU only, tensor per layer, ONE weight image  %.2f" % timed(graphed(lambda: step_like(HALVES, False, True, False))))


def single_distinct(strips=False):
    for i in range(LAYERS):
        K.conv3x3(acts[i], wpks[i], C, bias=b, relu=True, out=acts[i + 1], strips=strips)


print("  one chain, 3x48 tiles, tensor + weights per layer         %.2f" % timed(graphed(single_distinct)))
print("  three chains 6 + 5 + 5 images                             %.2f" % timed(graphed(lambda: multi([(0, 6), (6, 11), (11, 16)]))))
print("  three chains 6 + 5 + 5 images, alternating tables          %.2f" % timed(graphed(lambda: multi([(0, 6), (6, 11), (11, 16)], 0, True))))
print("  four quarter-batch chains, alternating tables              %.2f" % timed(graphed(lambda: multi([(0, 4), (4, 8), (8, 12), (12, 16)], 0, True))))


def epilogue_chain(parts, kind):
    """Two chains whose every link has the same epilogue: 'relu', 'mask', 'res0' or 'res2' (+res0 +res1); a tensor
    and a weight image per layer as in a training step."""
    cur = torch.cuda.current_stream()
    for k, rng in enumerate(parts):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for i in range(LAYERS):
                kw = {"relu": True} if kind == "relu" else {"mask": acts[(i + 3) % LAYERS]} if kind == "mask" else \
                     {"res0": acts[(i + 3) % LAYERS]} if kind == "res0" else \
                     {"res0": acts[(i + 3) % LAYERS], "res1": acts[(i + 7) % LAYERS]}
                K.conv3x3(acts[i], wpks[i], C, bias=b, out=acts[i + 1], images=rng, strips=2 if k else True,
                          plain_stores=kind != "relu", **kw)
    for k in range(len(parts)):
        cur.wait_stream(streams[k])


for kind in ("relu", "mask", "res0", "res2"):
    print("  two chains, every link with epilogue %-5s                 %.2f" % (kind, timed(graphed(lambda: epilogue_chain(HALVES, kind)))))
