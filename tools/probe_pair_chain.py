#!/usr/bin/env python3
"""Round 5: both half-batch conv3x3 + ReLU chains in ONE launch, the two strip tiles of a CU owned by
ONE 640-thread workgroup (conv3x3_pair_chain_kernel, csrc/conv3x3_pair_chain.inc; a measurement kernel that exists only
in the measurement library: tools/build_diag.sh diag; tools/larva_diag.h).

Checked bit for bit against the same chains as 2 x `layers` strip launches, then timed like bench.py's `roofline` block
(captured graph, replay / layers) beside bench.py's own two-chain figure in the same process.  Gate (VERDICT): <= 12.0 us per
full-batch layer on the 40-link chain.
usage: probe_pair_chain.py [layers=40]
env:   PAIR_LOCK="0,2,3,4" (phase locks to try, in chunks), PAIR_PRIO="1:1,..." (compute-wave priorities A:B), PAIR_NAPS"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from larvanet_amd import hip_lib, kernels as K
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diag_lib

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
C, B, P = bench.CH, bench.BATCH, bench.PATCH
fn = diag_lib.load().larva_conv3x3_pair_chain_probe
x0, wpk, b, bufs, rms = bench.chain_operands(dev, C, layers, False)
half = B // 2
parts = ((0, half), (half, B))
tabs = [K.strip_tile_table(P, P, dev, phase) for phase in (0, 1)]
tiles = tabs[0][1]
nwg = half * tiles


def neighbours(host_tab, n):
    """Per slot: the slots whose layer-L outputs the tile's layer-(L + 1) halo reads, itself included (a symmetric
    relation: the same list says whose inbox the tile bumps when it has finished a layer)."""
    geo = []
    for t in range(n):
        e = host_tab[t] & 0xffffffff
        y0, x0, rows = e & 0xfff, (e >> 12) & 0xfff, 5 if e >> 31 else 4
        geo.append((y0, y0 + rows, x0))
    out = -np.ones((n, 16), dtype=np.int32)
    deg = np.zeros(n, dtype=np.int32)
    for t, (a0, a1, ax) in enumerate(geo):
        k = 0
        for u, (b0, b1, bx) in enumerate(geo):
            if abs(ax - bx) <= 16 and b0 < a1 + 1 and b1 > a0 - 1:
                out[t, k] = u
                k += 1
        deg[t] = k
    return out, deg


nb = [neighbours(tabs[ph][2], tiles) for ph in (0, 1)]
nbr_dev = torch.from_numpy(np.stack([nb[0][0], nb[1][0]])).to(dev)
deg_dev = torch.from_numpy(np.stack([nb[0][1], nb[1][1]])).to(dev)
print("neighbours per tile (itself included): phase 0 %d-%d, phase 1 %d-%d" % (nb[0][1].min(), nb[0][1].max(), nb[1][1].min(), nb[1][1].max()))
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
state = torch.zeros((2 * half * tiles + 1) * 32, device=dev, dtype=torch.int32)   # a 128-byte line per inbox
xcc = torch.full((nwg,), -1, device=dev, dtype=torch.int32)
trace = torch.zeros(nwg * 2 * layers * 8, device=dev, dtype=torch.int64)
NAPS = int(os.environ.get("PAIR_NAPS", "2"))


def launch(lock, prio, with_trace=False):
    code = fn(pa.data_ptr(), pb.data_ptr(), wpk.data_ptr(), b.data_ptr(), half, P, P, P, tabs[0][0].data_ptr(), tabs[1][0].data_ptr(), nbr_dev.data_ptr(), deg_dev.data_ptr(), tiles,
              state.data_ptr(), xcc.data_ptr() if with_trace else None, trace.data_ptr() if with_trace else None, layers, lock, NAPS,
              prio[0], prio[1], torch.cuda.current_stream().cuda_stream)
    hip_lib.check(code, "larva_conv3x3_pair_chain_probe")


locks = [int(v) for v in os.environ.get("PAIR_LOCK", "0,2,3,4").split(",")]
prios = [tuple(int(u) for u in v.split(":")) for v in os.environ.get("PAIR_PRIO", "1:1").split(",")]
flop = bench.conv_flop(C)
for prio in prios:
    for lock in locks:
        pa.copy_(x0)
        pb.zero_()
        trace.zero_()
        torch.cuda.synchronize()
        launch(lock, prio, True)
        torch.cuda.synchronize()
        got = (pa, pb)[layers & 1]
        st = state.cpu()[::32].tolist()
        same = bool(torch.equal(got, ref))
        print("\n=== phase lock %d chunks, compute-wave priorities A:B = %d:%d" % (lock, prio[0], prio[1]))
        print("one launch vs %d strip launches: bit-identical %s (max |diff| %.3g), gave up waiting: %d, inbox / neighbours %s"
              % (2 * layers, same, float((got - ref).abs().max()), st[2 * half * tiles],
                 np.unique(np.asarray(st[:2 * half * tiles]).reshape(2, half, tiles) / np.stack([nb[0][1], nb[1][1]])[:, None, :]).tolist()))
        v = xcc.cpu().tolist()
        per_img = {}
        for e in v:
            per_img.setdefault(e >> 8, set()).add(e & 0xff)
        spread = [n for n, s in per_img.items() if len(s) != 1]
        print("  every image's %d workgroups on one XCD: %s" % (tiles, not spread))
        t = trace.cpu().numpy().reshape(nwg, 2, layers, 8).astype(np.float64) / 100.0   # us
        t = t[np.argsort(np.asarray(v) >> 8, kind="stable")]   # rows grouped by image (block -> tile goes through xcd_remap)
        t0 = t[:, :, 0, 0].min()
        for g in range(2):
            k_loop = np.median(t[:, g, :, 1] - t[:, g, :, 0])
            drain = np.median(t[:, g, :, 2] - t[:, g, :, 1])
            sig = np.median(t[:, g, :, 3] - t[:, g, :, 2])
            gap = np.median(t[:, g, 1:, 0] - t[:, g, :-1, 3]) if layers > 1 else 0.0
            per = np.median(t[:, g, 1:, 0] - t[:, g, :-1, 0]) if layers > 1 else 0.0
            print("  group %s (medians, us): first chunk ready -> K loop done %.2f | stores issued + drained %.2f | signalled %.2f | -> next layer's "
                  "first chunk ready %.2f | layer period %.2f" % ("AB"[g], k_loop, drain, sig, gap, per))
            if layers > 2:
                a, z = 2, layers - 1   # (steady layers)
                print("     layer edge: the tile's own signal -> its inbox is full %.2f us (loader reached the layer %.2f us before the signal) | "
                      "chunk 0 published +%.2f | compute sees it +%.2f | all 36 / 30 input pieces issued %.2f us after the inbox"
                      % (np.median(t[:, g, a:z, 5] - t[:, g, a - 1:z - 1, 3]), np.median(t[:, g, a - 1:z - 1, 3] - t[:, g, a:z, 4]),
                         np.median(t[:, g, a:z, 7] - t[:, g, a:z, 5]), np.median(t[:, g, a:z, 0] - t[:, g, a:z, 7]),
                         np.median(t[:, g, a:z, 6] - t[:, g, a:z, 5])))
        if os.environ.get("PAIR_DETAIL") and layers > 2:
            L = layers // 2
            for g in range(2):
                rows = t[:tiles, g]   # image 0 of the group
                base = rows[:, L, 5].min()
                print("  group %s, image 0, layer %d, per tile (us after the image's first 'counter there'): loader w-issued/counter/in-issued/published | "
                      "compute first chunk/K done/drained/signalled | next layer's counter" % ("AB"[g], L))
                for r in rows:
                    print("    " + " ".join("%6.2f" % (r[L, i] - base) for i in (4, 5, 6, 7)) + " | " +
                          " ".join("%6.2f" % (r[L, i] - base) for i in (0, 1, 2, 3)) + " | %6.2f" % (r[L + 1, 5] - base))
        if layers > 2:
            L = layers // 2
            off = np.median(t[:, 1, L, 0] - t[:, 0, L, 0])
            print("  layer %d: group B starts its K loop %.2f us after group A (medians over the workgroups); whole launch %.1f us = %.2f us per layer"
                  % (L, off, t[:, :, -1, 3].max() - t0, (t[:, :, -1, 3].max() - t0) / layers))
        if not same:
            continue
        pa.copy_(x0)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch(lock, prio)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            launch(lock, prio)
        ms = bench.replay_ms(graph, 10)
        print("  ONE launch, both half batches: %.2f us per full-batch layer = %.3f of the fp32 matrix peak  (gave up: %d)"
              % (ms * 1e3 / layers, flop / (ms / layers * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, int(state[2 * half * tiles * 32])))
if layers == bench.CHAIN_SHORT:
    slope, _, per = bench.dual_chain_time_ms(dev, C)
    print("\n%d launches per half batch: %.2f us per full-batch layer = %.3f (bench.py's roofline.avg_ms; steady state %.2f us)"
          % (layers, per * 1e3, flop / (per * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS, slope * 1e3))
