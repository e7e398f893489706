#!/usr/bin/env python3
"""One full-image 48 -> 48 conv3x3 layer (1 x 48 x 339 x 510, pitch 512: what validate.py's forward issues 34 times per
image), back to back in a captured graph of 20 launches: persistent tiles against one workgroup per tile
(LARVA_PERSIST=0), per epilogue.   usage: bench_wide_layer.py [channels=48]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from larvanet_amd import kernels as K

C = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0)
H, W, P = 339, 510, 512
g = torch.Generator().manual_seed(0)
x = torch.zeros(1, C, H, P, device=dev)
x[..., :W] = (torch.randn(1, C, H, W, generator=g) * 20).to(dev)
r0, r1 = x.flip(1).contiguous(), x.flip(2).contiguous()
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
b = torch.zeros(C, device=dev)
fwd, _ = K.pack_weights(w)
bufs = [torch.empty_like(x) for _ in range(2)]
flop = 2 * 9 * C * C * H * W
kinds = {"relu": dict(relu=True), "res1": dict(res0=r0), "res2": dict(res0=r0, res1=r1)}
for mode in ("1", "0"):
    os.environ["LARVA_PERSIST"] = mode
    row = []
    for name, kw in kinds.items():
        def chain():
            src = x
            for i in range(20):
                K.conv3x3(src, fwd, C, bias=b, out=bufs[i & 1], logical_w=W, tile_rows=3, **kw)
                src = bufs[i & 1]
        chain()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            chain()
        ms = bench.replay_ms(graph, 10)
        row.append("%s %.1f us (%.3f)" % (name, ms * 1e3 / 20, flop / (ms / 20 * 1e-3) / 1e12 / bench.FP32_MFMA_PEAK_TFLOPS))
    print("%s: %s" % ("persistent tiles       " if mode == "1" else "one workgroup per tile ", "   ".join(row)))
