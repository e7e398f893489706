"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.

BASELINE.json configs[0], end to end: `EDSR-baseline x4, batch 4, 48x48 LR patches on PyTorch-CPU via train.py` --
the reference's original driver loop (train.py:19-113: three-stage flag chaining, model.prepare, arguments.json,
`while model.global_step < max_steps`: get_next_train_scale -> dataloader.get_patch_batch -> as_tensor ->
model.train_step(input_list, scale, truth_list, summary) -> log every log_freq -> save every save_freq) restated over
oracle/edsr_torch.py (the reference's models/edsr.py plugin, pinned by fixture F15) and any loader with the
reference's loader surface (default: the dataset-free synthetic one).  CPU only, no HIP kernels: EDSR is not on the
hot path; this is the plumbing the config names, and what bench.py's `cpu_baseline.edsr_train_step` times.

  python -m oracle.train_edsr_cpu --max_steps=20 --train_path=/tmp/edsr
"""
import argparse
import copy
import importlib
import json
import os
import time

import numpy as np
import torch

from . import edsr_torch as E


class EDSR:
    """models/edsr.py:18-125: the plugin wrapper (parse_args / prepare / train_step / upscale / save / restore)."""

    def parse_args(self, args):
        p = argparse.ArgumentParser()
        p.add_argument("--edsr_conv_features", type=int, default=64)
        p.add_argument("--edsr_res_blocks", type=int, default=16)
        p.add_argument("--edsr_res_weight", type=float, default=1.0)
        p.add_argument("--edsr_learning_rate", type=float, default=1e-4)
        p.add_argument("--edsr_learning_rate_decay", type=float, default=0.5)
        p.add_argument("--edsr_learning_rate_decay_steps", type=int, default=200000)
        self.args, remaining = p.parse_known_args(args=args)
        return copy.deepcopy(self.args), remaining

    def prepare(self, is_training, scales, global_step=0):
        for scale in scales:
            if scale not in (2, 3, 4):
                raise ValueError("Unsupported scale is provided.")
        if len(scales) != 1:
            raise ValueError("Only one scale should be provided.")
        self.scale_list, self.scale = scales, scales[0]
        self.device = torch.device("cpu")
        a = self.args
        self.sd = E.init_state_dict(a.edsr_conv_features, a.edsr_res_blocks, self.scale)
        self._step = E.make_trainer(self.sd, a.edsr_res_blocks, self.scale, lr=a.edsr_learning_rate,
                                    lr_decay=a.edsr_learning_rate_decay, lr_decay_steps=a.edsr_learning_rate_decay_steps,
                                    res_weight=a.edsr_res_weight, global_step=global_step) if is_training else None

    @property
    def global_step(self):
        return self._step.state["global_step"]

    def get_next_train_scale(self):
        return self.scale_list[np.random.randint(len(self.scale_list))]

    def get_lr(self):
        a = self.args
        return a.edsr_learning_rate * (a.edsr_learning_rate_decay ** (self.global_step // a.edsr_learning_rate_decay_steps))

    def train_step(self, input_list, scale, truth_list, summary=None):
        return self._step(torch.as_tensor(np.asarray(input_list), dtype=torch.float32),
                          torch.as_tensor(np.asarray(truth_list), dtype=torch.float32))

    def state_dict(self):
        return {k: v.detach().clone() for k, v in (self._step.params if self._step else self.sd).items()}

    def upscale(self, input_list, scale):
        with torch.no_grad():
            x = torch.as_tensor(np.asarray(input_list), dtype=torch.float32)
            return E.forward(self.state_dict(), x, self.args.edsr_res_blocks, self.scale, self.args.edsr_res_weight).numpy()

    def save(self, base_path):
        path = os.path.join(base_path, "model_%d.pth" % self.global_step)   # models/edsr.py:60-62
        torch.save(self.state_dict(), path)
        return path


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--dataloader", type=str, default="synthetic_loader")
    p.add_argument("--batch_size", type=int, default=4)            # (the reference's default is 16; configs[0] says 4)
    p.add_argument("--input_patch_size", type=int, default=48)
    p.add_argument("--scales", type=str, default="4")
    p.add_argument("--train_path", type=str, default="runs/edsr_cpu")
    p.add_argument("--max_steps", type=int, default=300000)
    p.add_argument("--log_freq", type=int, default=10)
    p.add_argument("--save_freq", type=int, default=10000)
    p.add_argument("--global_step", type=int, default=0)
    p.add_argument("--threads", type=int, default=0, help="torch CPU threads (0: leave as is)")
    args, remaining = p.parse_known_args(argv)
    if args.threads > 0:
        torch.set_num_threads(args.threads)
    scale_list = [int(v) for v in args.scales.split(",")]
    os.makedirs(args.train_path, exist_ok=True)
    loader = importlib.import_module("larvanet_amd.dataloaders." + args.dataloader).create_loader()
    _, remaining = loader.parse_args(remaining)
    loader.prepare(scales=scale_list)
    model = EDSR()
    model_args, remaining = model.parse_args(remaining)
    model.prepare(is_training=True, scales=scale_list, global_step=args.global_step)
    if remaining:
        print("WARNING: found unhandled arguments: %s" % remaining)
    with open(os.path.join(args.train_path, "arguments.json"), "w") as f:
        f.write(json.dumps({**vars(args), **vars(model_args)}, sort_keys=True, indent=2))
    print("begin training")
    local_step, losses = 0, []
    while model.global_step < args.max_steps:
        global_train_step = model.global_step + 1
        local_step += 1
        t0 = time.time()
        scale = model.get_next_train_scale()
        input_list, truth_list = loader.get_patch_batch(batch_size=args.batch_size, scale=scale,
                                                        input_patch_size=args.input_patch_size)
        lr = model.get_lr()
        loss = model.train_step(input_list=input_list, scale=scale, truth_list=truth_list, summary=None)
        losses.append(loss)
        duration = time.time() - t0
        if local_step % args.log_freq == 0:
            print("step %d, lr %f, loss %.6f (%.3f sec/batch)" % (global_train_step, lr, loss, duration))
        if local_step % args.save_freq == 0:
            model.save(base_path=args.train_path)
            print("saved a model checkpoint at step %d" % global_train_step)
    print("finished")
    return model, losses


if __name__ == "__main__":
    main()
