#!/usr/bin/env python3
"""The register-staged weight-gradient kernel with one and with two workgroups per CU (32 layers x 8 / x 16 splits
+ the reduction, captured graph, HIP events): us per layer and fraction of the fp32 MFMA peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from larvanet_amd import kernels as K

dev = torch.device("cuda", 0)
for c in (32, 64):
    g = torch.Generator().manual_seed(6)
    dy = (torch.randn(16, c, 48, 48, generator=g) * 1e-3).to(dev)
    xs = (torch.randn(16, c, 48, 48, generator=g) * 20).to(dev)
    js = [{"dy": dy + 0, "x": xs + 0, "dw": torch.empty(c, c, 3, 3, device=dev), "db": torch.empty(c, device=dev)} for _ in range(32)]
    for splits in (8, 12, 16, 24):
        K.conv3x3_wgrad(js, c, c, splits)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            K.conv3x3_wgrad(js, c, c, splits)
        ms = bench.replay_ms(graph, 10)
        tf = bench.conv_flop(c) * 32 / (ms * 1e-3) / 1e12
        print("c=%d  32 layers x %2d splits (share %d): %.2f us per layer  %.1f TFLOP/s = %.3f of peak"
              % (c, splits, K.wgrad_cu_share(c, c), ms * 1e3 / 32, tf, tf / bench.FP32_MFMA_PEAK_TFLOPS))
