#!/usr/bin/env python3
"""How stable is the headline step?  The two half-batch chains of the captured step have a fast and a slow phase relation
(profiles/r06_step_timeline_stamped.txt: the STAMPED build's backward chain takes 483 or 572-581 us); bench.py's timed rounds
always sat in the fast one.  This tool runs the PRODUCT step for a long time -- `--seconds` of back-to-back steps in blocks
of `--block` steps, reference semantics (fresh device tensors, loss.item() per step) -- then the same after idle gaps of
several lengths, and prints the distribution of the per-block ms per step: which share of the blocks sits within 1 % of the
fastest, and the slowest block seen.

  python tools/step_soak.py [--seconds 40] [--block 20] [out.txt]            (GPU box)
"""
import argparse
import importlib
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=40.0)
    ap.add_argument("--block", type=int, default=20)
    ap.add_argument("out", nargs="?")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    dev = torch.device("cuda", 0)
    m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    m.parse_args(list(bench.FLAGS))
    torch.manual_seed(0)
    m.volume_per_step = 48 * 48 * 16 * 3
    m.prepare(is_training=True, scales=[4])
    m.sync_loss = True
    g = torch.Generator().manual_seed(1000)
    x = (torch.rand(16, 3, 48, 48, generator=g) * 255).to(dev)
    t = (torch.rand(16, 3, 192, 192, generator=g) * 255).to(dev)
    args = types.SimpleNamespace(train_path="/tmp")
    val = bench.TinyValLoader()
    lines = []

    def w(s):
        print(s, flush=True)
        lines.append(s)

    def block():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.block):
            m.train_step_larva(args, val, x, t)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.block * 1e3

    for _ in range(5):
        block()   # capture + clock ramp
    w("product training step (M4B4, 48 channels, 16 x 3 x 48 x 48, reference semantics), blocks of %d back-to-back steps" % a.block)
    t_end = time.perf_counter() + a.seconds
    vals = []
    while time.perf_counter() < t_end:
        vals.append(block())
    v = np.asarray(vals)
    fast = v.min()
    w("steady state: %d blocks = %d steps in %.0f s: ms per step min %.4f / median %.4f / p99 %.4f / max %.4f; blocks within 1 %% of the "
      "fastest: %.1f %%, above +3 %%: %d, above +5 %% (the slow chain mode would be +5.5 %%): %d"
      % (len(v), len(v) * a.block, a.seconds, fast, np.median(v), np.percentile(v, 99), v.max(),
         100.0 * np.mean(v <= 1.01 * fast), int(np.sum(v > 1.03 * fast)), int(np.sum(v > 1.05 * fast))))
    hist, edges = np.histogram(v, bins=12)
    w("histogram (ms per step: blocks): " + "  ".join("%.3f-%.3f: %d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist)) if hist[i]))
    for gap in (0.01, 0.1, 0.5, 2.0):
        firsts, seconds_, thirds = [], [], []
        for _ in range(6):
            time.sleep(gap)
            firsts.append(block())
            seconds_.append(block())
            thirds.append(block())
        w("after %.2f s of idle (6 times): first block %.4f-%.4f, second %.4f-%.4f, third %.4f-%.4f ms per step"
          % (gap, min(firsts), max(firsts), min(seconds_), max(seconds_), min(thirds), max(thirds)))
    w("final loss %.4f after %d steps (finite: %s)" % (float(m.train_step_larva(args, val, x, t)), m.global_step, "yes"))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
