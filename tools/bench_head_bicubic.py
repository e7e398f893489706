#!/usr/bin/env python3
"""The two HBM-bound launches of a full-image forward (3 x 339 x 510 LR image -> 33 MB each): the 3 -> 48 head as the padded
MFMA launch against the direct kernel, and the bicubic x4 base image; at the training size too.  Captured graphs of 20 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
w = (torch.randn(48, 3, 3, 3, generator=g) * 0.1).to(dev)
b = (torch.randn(48, generator=g)).to(dev)
w16 = torch.zeros(48, 16, 3, 3, device=dev)
w16[:, :3] = w
fwd16, _ = K.pack_weights(w16)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for _ in range(20):
            fn()
    return bench.replay_ms(graph, 10) * 1e3 / 20


for N, H, W in ((1, 339, 510), (16, 48, 48)):
    P = (W + 3) // 4 * 4
    x = (torch.rand(N, 3, H, W, generator=g) * 255).to(dev)
    x16 = torch.zeros(N, 16, H, P, device=dev)
    x16[:, :3, :, :W] = x
    out = torch.empty(N, 48, H, P, device=dev)
    mb = out.numel() * 4 / 1e6
    t_mfma = timed(lambda: K.conv3x3(x16, fwd16, 48, bias=b, out=out, logical_w=W))
    ref = out.clone()
    t_dir = timed(lambda: K.head_conv3_direct(x, w, b, pitch=P))
    d = float((K.head_conv3_direct(x, w, b, pitch=P) - ref).abs().max())
    t_bic = timed(lambda: K.bicubic4(x))
    print("%d x 3 x %d x %d: head (%.1f MB out) padded MFMA %.1f us = %.2f TB/s | direct %.1f us = %.2f TB/s (max |diff| %.2g) | "
          "bicubic x4 %.1f us = %.2f TB/s" % (N, H, W, mb, t_mfma, mb / t_mfma, t_dir, mb / t_dir, d, t_bic, N * 3 * 16 * H * W * 4 / 1e6 / t_bic))
