#!/usr/bin/env python3
"""bench.py's weight-gradient block (one launch over 40 layers + reduction, captured and replayed) at the given channel
counts.  LARVA_WGRAD_PIPE=0 in the environment gives the register-staged kernel of every shape for a same-box A/B.
usage: bench_wgrad_widths.py [C ...]   (default 32 48 64)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
for c in [int(a) for a in sys.argv[1:]] or [32, 48, 64]:
    b = bench.wgrad_block(dev, c)
    print("C=%d  %.2f us per layer  %.1f TFLOP/s = %.3f of peak  (%s)" % (c, b["ms_per_layer"] * 1e3, b["achieved"], b["frac"], b["kernel"]),
          flush=True)
