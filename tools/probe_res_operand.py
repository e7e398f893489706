#!/usr/bin/env python3
"""Where do the +3 / +6 us of a residual epilogue on a full-image layer come from?  The same persistent conv (1 x 48 x 339 x
510) with its residual operand(s) (a) in tensors of their own (cold: 33 MB more to fetch each), (b) = the layer's own INPUT
tensor (the lines the loader wave has just streamed), (c) = one 1-row tensor every tile reads again (hot in every L2).
  python tools/probe_res_operand.py        (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from larvanet_amd import kernels as K

C = 48
dev = torch.device("cuda", 0)
H, W, P = 339, 510, 512
g = torch.Generator().manual_seed(0)
x = torch.zeros(1, C, H, P, device=dev)
x[..., :W] = (torch.randn(1, C, H, W, generator=g) * 20).to(dev)
r0, r1 = x.flip(1).contiguous(), x.flip(2).contiguous()
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
b = torch.zeros(C, device=dev)
fwd, _ = K.pack_weights(w)
bufs = [torch.empty_like(x) for _ in range(3)]
flop = 2 * 9 * C * C * H * W


def timed(kw_of):
    def chain():
        src = x
        for i in range(20):
            K.conv3x3(src, fwd, C, bias=b, out=bufs[i % 3], logical_w=W, tile_rows=3, **kw_of(src, bufs[(i + 2) % 3]))
            src = bufs[i % 3]
    chain()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        chain()
    return bench.replay_ms(graph, 10) * 1e3 / 20


print("conv + ReLU                                   %.1f us" % timed(lambda src, prev: dict(relu=True)))
print("+ res0, a tensor of its own (cold)            %.1f us" % timed(lambda src, prev: dict(res0=r0)))
print("+ res0 = the layer's own input                %.1f us" % timed(lambda src, prev: dict(res0=src)))
print("+ res0 = the previous layer's input (as a residual block has it)  %.1f us" % timed(lambda src, prev: dict(res0=prev)))
print("+ res0 + res1, tensors of their own (cold)    %.1f us" % timed(lambda src, prev: dict(res0=r0, res1=r1)))
print("+ res0 = res1 = the layer's own input         %.1f us" % timed(lambda src, prev: dict(res0=src, res1=src)))
