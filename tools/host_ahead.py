#!/usr/bin/env python3
"""Is the host ahead of the GPU in the bench loop?  Per step: host time spent inside train_step_larva
(issue), and the time until the GPU has finished everything (20 steps)."""
import importlib
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device("cuda", 0)
model = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
model.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
torch.manual_seed(0)
model.volume_per_step = 1
model.prepare(is_training=True, scales=[4])
model.sync_loss = False
g = torch.Generator().manual_seed(1)
x = (torch.rand(16, 3, 48, 48, generator=g) * 255).to(dev)
t = (torch.rand(16, 3, 192, 192, generator=g) * 255).to(dev)


class Val:
    def get_num_images(self):
        return 1

    def get_image_pair(self, image_index, scale):
        rng = np.random.RandomState(3)
        return (rng.randint(0, 256, (3, 24, 24)).astype(np.float32), rng.randint(0, 256, (3, 96, 96)).astype(np.float32), "s")


args = types.SimpleNamespace(train_path="/tmp")
for _ in range(5):
    model.train_step_larva(args, Val(), x, t)
x, t = model.input_buffers(x.shape, t.shape)
torch.cuda.synchronize()
for label in ("as is",):
    issue = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter()
        model.train_step_larva(args, Val(), x, t)
        issue.append(time.perf_counter() - a)
    t_issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    a = time.perf_counter()
    model._graph.replay()
    r = time.perf_counter() - a
    torch.cuda.synchronize()
    print("dual_chain=%s: host issue per step median %.0f us (min %.0f max %.0f), all 20 issued after %.2f ms, GPU done after %.2f ms "
          "(%.3f ms/step); one graph.replay() call %.0f us"
          % (model.dual_chain, np.median(issue) * 1e6, min(issue) * 1e6, max(issue) * 1e6, t_issued * 1e3, total * 1e3, total / 20 * 1e3, r * 1e6))
