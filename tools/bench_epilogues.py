#!/usr/bin/env python3
"""The strip kernel's epilogue variants at the training shape (8 x 48 x 48 x 48 per launch), two ways:
  alone     one half-batch launch running alone, kernel-attached events (what rocprofv3 reports per dispatch)
  chains    two half-batch chains of 40 links that ALL have that epilogue, a tensor and a weight image per layer
            as in a training step, captured graph, per full-batch layer
Run with LARVA_HIP_LIB=<variant .so> (and LARVA_DIAG_LIB for the `alone` column) for a same-box A/B (tools/ab_lib.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from larvanet_amd import kernels as K
import diag_lib   # (tools/diag_lib.py)

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
N, LAYERS = 16, 40
C = int(sys.argv[1]) if len(sys.argv) > 1 else 48
b = torch.zeros(C, device=dev)
acts = [(torch.randn(N, C, 48, 48, generator=g) * 20).to(dev)] + [torch.empty(N, C, 48, 48, device=dev) for _ in range(LAYERS)]
for a in acts[1:]:
    a.copy_(acts[0])
wpks = [K.pack_weights((torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev))[0] for _ in range(LAYERS)]
K.strip_tile_table(48, 48, dev, 0)
K.strip_tile_table(48, 48, dev, 1)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
HALVES = [(0, 8), (8, 16)]


def operands(kind, i):
    if kind == "relu":
        return {"relu": True}
    if kind == "plain":         # no epilogue operand, non-temporal stores: the floor of the mask / residual links
        return {}
    if kind == "mask":
        return {"mask": acts[(i + 3) % LAYERS]}
    if kind == "res0":
        return {"res0": acts[(i + 3) % LAYERS]}
    return {"res0": acts[(i + 3) % LAYERS], "res1": acts[(i + 7) % LAYERS]}


def epilogue_chain(kind):
    cur = torch.cuda.current_stream()
    for k, rng in enumerate(HALVES):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            for i in range(LAYERS):
                K.conv3x3(acts[i], wpks[i], C, bias=b, out=acts[i + 1], images=rng, strips=2 if k else True,
                          plain_stores=kind.startswith("relu"), **operands(kind, i))
    for k in range(2):
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


print("lib: %s   channels: %d" % (os.environ.get("LARVA_HIP_LIB", "default"), C))
for kind in ("relu", "plain", "mask", "res0", "res2"):
    kw = operands(kind, 0)
    if diag_lib.available():    # kernel-attached timing lives in the measurement library (tools/build_diag.sh)
        diag_lib.conv3x3_strips_timed(acts[0], wpks[0], C, b, acts[1], 5, images=(0, 8), **kw)
        mean, best = diag_lib.conv3x3_strips_timed(acts[0], wpks[0], C, b, acts[1], 100, images=(0, 8), **kw)
    else:
        mean = best = float("nan")
    for a in acts[1:]:
        a.copy_(acts[0])
    chain = timed(graphed(lambda: epilogue_chain(kind)))
    print("  %-9s  alone: mean %.2f us  min %.2f us     two chains, every link: %.2f us per full-batch layer" % (kind, mean * 1e3, best * 1e3, chain))
