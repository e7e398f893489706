#!/usr/bin/env python3
"""What would the layer chains gain from replacing kernel boundaries by in-launch dependencies?  

conv3x3_strip_chain_kernel (larva_conv3x3_chain_probe, a measurement kernel) runs a whole chain of conv3x3 + ReLU layers
of one half batch in ONE launch: every workgroup keeps its strip tile, a layer's input pieces wait on a per-image counter
that the image's 32 workgroups bump after draining their stores, LDS-DMA loads bypass the vector L1.  Two such launches
(the two half batches, 256 workgroups each, the two tile-table phases) on two streams fill the chip exactly.

Checked bit for bit against the same chain as 2 x `layers` strip launches, then timed like bench.py's `roofline` block
(captured graph, replay / layers) beside bench.py's own two-chain figure in the same process.
usage: probe_chain_kernel.py [layers=40]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from larvanet_amd import hip_lib, kernels as K

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
C, B, P = bench.CH, bench.BATCH, bench.PATCH
lib = hip_lib.load()
x0, wpk, b, bufs, rms = bench.chain_operands(dev, C, layers, False)
half = B // 2
parts = ((0, half), (half, B))
tabs = [K.strip_tile_table(P, P, dev, phase) for phase in (0, 1)]
if os.environ.get("CHAIN_PROBE_SAME_PHASE"):   # both half batches cut their images the same way (timing only: the reference
    tabs[1] = tabs[0]                          # chain uses the two phases, so the bit-for-bit line then fails for half 1)
tiles = tabs[0][1]
print("%d layers of 16 x %d x %d x %d, %d strip tiles per image, activations RMS %.1f" % (layers, C, P, P, tiles, float(x0.pow(2).mean().sqrt())))

# reference: the chain as strip launches (one stream)
ra, rb = x0.clone(), torch.empty_like(x0)
rbuf = [ra, rb]
for L in range(layers):
    for k in range(2):
        K.conv3x3(rbuf[L & 1], wpk, C, bias=b, relu=True, out=rbuf[(L + 1) & 1], images=parts[k], strips=2 if k else True, plain_stores=True)
torch.cuda.synchronize()
ref = rbuf[layers & 1]

pa, pb = x0.clone(), torch.zeros_like(x0)
state = [torch.zeros(half + 1, device=dev, dtype=torch.int32) for _ in range(2)]
xcc = [torch.full((half * tiles,), -1, device=dev, dtype=torch.int32) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
img_floats = C * P * P


trace = [torch.zeros(half * tiles * layers * 8, device=dev, dtype=torch.int64) for _ in range(2)]


def launch(k, st, with_xcc=False):
    off = parts[k][0] * img_floats * 4
    code = lib.larva_conv3x3_chain_probe(pa.data_ptr() + off, pb.data_ptr() + off, wpk.data_ptr(), b.data_ptr(), half, P, P, P,
                                         tabs[k][0].data_ptr(), tiles, state[k].data_ptr(), xcc[k].data_ptr() if with_xcc else None,
                                         trace[k].data_ptr() if with_xcc else None, layers, NAPS, PLAIN, st.cuda_stream)
    hip_lib.check(code, "larva_conv3x3_chain_probe")


NAPS = int(os.environ.get("CHAIN_PROBE_NAPS", "2"))   # 64-clock naps between two looks at an image's counter
PLAIN = int(os.environ.get("CHAIN_PROBE_PLAIN", "1"))  # 0: non-temporal output stores
STAGGER = float(os.environ.get("CHAIN_PROBE_STAGGER_US", "0"))   # a sleeping launch of this many us in front of half 1's launch
ONLY = os.environ.get("CHAIN_PROBE_ONLY")   # "0" / "1": launch one half batch only (timing; the comparison then fails)


def both(with_xcc=False):
    cur = torch.cuda.current_stream()
    for k, st in enumerate(streams):
        if ONLY is not None and int(ONLY) != k:
            continue
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            if k == 1 and STAGGER > 0:
                hip_lib.check(lib.larva_delay_ticks(int(STAGGER * 100), st.cuda_stream), "larva_delay_ticks")
            launch(k, st, with_xcc)
    for st in streams:
        cur.wait_stream(st)


both()            # (first call: one-off host work between the two launches; the checked, stamped run follows)
torch.cuda.synchronize()
pa.copy_(x0)
torch.cuda.synchronize()
both(True)
torch.cuda.synchronize()
got = (pa, pb)[layers & 1]
gave_up = [int(s[half]) for s in state]
same = bool(torch.equal(got, ref))
diff = float((got - ref).abs().max())
print("one launch per half batch vs %d strip launches: bit-identical %s (max |diff| %.3g), gave up waiting: %s, counters %s"
      % (2 * layers, same, diff, gave_up, [s[:half].tolist() for s in state]))
for k in range(2):
    v = xcc[k].cpu().tolist()
    per_img = {}
    for e in v:
        per_img.setdefault(e >> 8, set()).add(e & 0xff)
    print("  half %d: XCDs per image %s" % (k, {n: sorted(s) for n, s in sorted(per_img.items())}))

import numpy as np
origin = min(float(trace[k].cpu().numpy().reshape(half * tiles, layers, 8)[:, 0, 0].min()) for k in range(2)
             if ONLY is None or int(ONLY) == k) / 100.0
for k in range(2):
    if ONLY is not None and int(ONLY) != k:
        continue
    t = trace[k].cpu().numpy().reshape(half * tiles, layers, 8).astype(np.float64) / 100.0   # us
    t0 = t[:, 0, 0].min()
    print("  half %d, medians over its %d workgroups (us): layer entered (after the first entry) | input released after | stores issued after | barrier after | next layer's entry"
          % (k, half * tiles))
    for L in sorted(set([0, 1, 2, 3, layers // 2, layers - 2, layers - 1])):
        if not 0 <= L < layers:
            continue
        ent = t[:, L, 0]
        nxt = t[:, L + 1, 0] if L + 1 < layers else t[:, L, 3]
        print("    layer %3d: %8.2f (first %8.2f last %8.2f) | %5.2f | %5.2f | %5.2f | %5.2f"
              % (L, np.median(ent) - t0, ent.min() - t0, ent.max() - t0, np.median(t[:, L, 1] - ent) if L else 0.0,
                 np.median(t[:, L, 2] - ent), np.median(t[:, L, 3] - ent), np.median(nxt - ent)))
    if os.environ.get("CHAIN_PROBE_DETAIL"):
        L = layers // 2
        xs = xcc[k].cpu().numpy()
        img = xs >> 8
        for n_ in (0, 1):
            sel = np.where(img == n_)[0]
            rel = t[sel, L, 1].min()
            rows = ["%5.1f/%5.1f/%5.1f/%5.1f/%5.1f/%5.1f/%5.1f" % tuple(t[w, L, i] - rel for i in (0, 1, 2, 4, 6, 5, 3)) for w in sel]
            print("    layer %d, image %d (XCD %d), per workgroup entered/released/wave 0 stores issued/wave 0 drained/wave 3 drained/loader done/barrier (us after the image's first release):" % (L, n_, xs[sel[0]] & 0xff))
            for i in range(0, len(rows), 2):
                print("      " + "   ".join(rows[i:i + 2]))
    print("    whole chain %.1f us = %.2f us per half-batch layer; it ran from %.1f to %.1f us after the first workgroup of either launch"
          % (t[:, -1, 3].max() - t0, (t[:, -1, 3].max() - t0) / layers, t0 - origin, t[:, -1, 3].max() - origin))

pa.copy_(x0)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    both()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode="thread_local"):
    both()
t = bench.replay_ms(graph, 10)
flop = bench.conv_flop(C) * (0.5 if ONLY is not None else 1.0)
print("ONE launch per half batch:  %.2f us per %s layer = %.3f of the fp32 matrix peak  (gave up: %s)"
      % (t * 1e3 / layers, "HALF-batch (one launch running alone)" if ONLY is not None else "full-batch",
         flop / (t / layers * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, [int(s[half]) for s in state]))
if layers == bench.CHAIN_SHORT:
    slope, _, per = bench.dual_chain_time_ms(dev, C)
    print("%d launches per half batch: %.2f us per full-batch layer = %.3f (bench.py's roofline.avg_ms; steady state %.2f us)"
          % (layers, per * 1e3, flop / (per * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, slope * 1e3))
