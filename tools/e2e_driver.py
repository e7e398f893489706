#!/usr/bin/env python3
"""End-to-end throughput of the DRIVER loop (train_larva.py counterpart, device-resident patch loader,
M4B4, batch 16 x 48 x 48) -- loader + hand-over + train_step_larva per step, not just the step bench.py times."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from larvanet_amd import train_larva

steps = int(os.environ.get("STEPS", "400"))
for flags, label in ((["--async_loss"], "async loss"), ([], "loss.item() every step (reference behaviour)")):
    with tempfile.TemporaryDirectory() as d:
        argv = ["--model=LarvaNet", "--dataloader=device_patch_loader", "--device_source=synthetic_loader",
                "--val_dataloader=synthetic_loader", "--train_path", d, "--batch_size=16", "--input_patch_size=48",
                "--num_modules=4", "--num_blocks=4,4,4,4", "--synthetic_images=32", "--synthetic_lr_size=96",
                "--data_seed=1", "--log_freq=100000"] + flags
        train_larva.main(argv + ["--max_steps=60"])       # warm-up: capture, allocator
        times = []
        for n in (steps // 4, steps):                       # two run lengths: the slope is the loop itself
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train_larva.main(argv + ["--max_steps=%d" % n])
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    per_step = (times[1] - times[0]) / (steps - steps // 4)
    print("DRIVER LOOP, %s: %.3f ms per step (slope between %d and %d steps; %.3f ms incl. model construction and capture)"
          % (label, per_step * 1e3, steps // 4, steps, times[1] / steps * 1e3), flush=True)
