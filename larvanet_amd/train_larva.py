"""Training driver: counterpart of the reference's train_larva.py for the MI355X plugins.

    python -m larvanet_amd.train_larva --model=LarvaNet --num_modules=4 --num_blocks=4,4,4,4 \\
        --dataloader=div2k_train_loader --data_input_path=... --data_truth_path=... --train_path=runs/x4
    torchrun --nproc-per-node 8 -m larvanet_amd.train_larva ...        (data parallel over RCCL)

Same flags, same three-stage argument chaining (driver -> loader -> model, leftovers only warn),
same loop (scale -> batch -> as_tensor -> train_step_larva), same arguments.json dump.  Added:
--max_steps actually ends the loop, one process per GPU with per-rank patch streams, and
volume_per_step counts the GLOBAL batch so that "volume" keeps its meaning under data parallelism.
"""
import argparse
import importlib
import json
import os
import time

import numpy as np
import torch

from . import dist as ldist


class _NullSummary:
    """Stand-in when tensorboard is not installed."""

    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass

    def close(self):
        pass


def make_summary_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=path)
    except Exception:
        return _NullSummary()


def round_to_1(x):
    """train_larvaV2.py:15-16: x rounded to one significant digit."""
    from math import floor, log10
    return round(x, -int(floor(log10(abs(x)))))


def build_parser(v2=False):
    p = argparse.ArgumentParser()
    if v2:
        p.add_argument("--steps_per_epoch", type=float, help="Num of steps on 1 epoch.")
    p.add_argument("--dataloader", type=str, default="combined_loader")
    p.add_argument("--val_dataloader", type=str, default="div2k_val_loader")
    p.add_argument("--model", type=str, default="LarvaNet")
    p.add_argument("--batch_size", type=int, default=16, help="patches per step PER GPU")
    p.add_argument("--input_patch_size", type=int, default=48)
    p.add_argument("--scales", type=str, default="4")
    p.add_argument("--cuda_device", type=str, default=None,
                   help="device list for HIP_VISIBLE_DEVICES (single process only; torchrun sets LOCAL_RANK)")
    p.add_argument("--train_path", type=str, default="runs/larvanet")
    p.add_argument("--max_steps", type=int, default=300000)
    p.add_argument("--log_freq", type=int, default=10)
    p.add_argument("--summary_freq", type=int, default=1000)
    p.add_argument("--save_freq", type=int, default=10000)
    p.add_argument("--sleep_ratio", type=float, default=0.0,
                   help="idle fraction per step (the reference defaults to 0.05 'to prevent overheating')")
    p.add_argument("--restore_path", type=str)
    p.add_argument("--restore_target", type=str)
    p.add_argument("--global_step", type=int, default=0)
    p.add_argument("--async_loss", action="store_true", help="do not read the loss back every step")
    p.add_argument("--allow_eager_fallback", action="store_true",
                   help="if a hipGraph capture fails, go on with one launch per kernel (~2.4x slower) instead of raising")
    p.add_argument("--save_train_state", action="store_true",
                   help="also save optimizer / scheduler / counters / RNG with every checkpoint")
    p.add_argument("--resume_state", type=str, help="train_state_step*.pth to continue from (with --restore_path)")
    return p


def main(argv=None, v2=False):
    """v2: the reference's train_larvaV2.py variant of the same loop (see larvanet_amd/train_larvaV2.py)."""
    args, remaining = build_parser(v2).parse_known_args(argv)
    if args.cuda_device is not None and "LOCAL_RANK" not in os.environ:
        os.environ["HIP_VISIBLE_DEVICES"] = args.cuda_device  # before the first device use
    rank, world = ldist.init_from_env()
    ldist.limit_host_threads()   # this rank's share of the host's cores (LOCAL_WORLD_SIZE ranks per node)
    scales = [int(s) for s in args.scales.split(",")]
    os.makedirs(args.train_path, exist_ok=True)

    print("prepare data loader - %s" % args.dataloader)
    loader = importlib.import_module("larvanet_amd.dataloaders." + args.dataloader).create_loader()
    loader_args, remaining = loader.parse_args(remaining)
    loader.prepare(scales=scales)
    val_loader = importlib.import_module("larvanet_amd.dataloaders." + args.val_dataloader).create_loader()
    val_args, remaining = val_loader.parse_args(remaining)
    val_loader.prepare(scales=scales)

    print("prepare model - %s" % args.model)
    model = importlib.import_module("larvanet_amd.models." + args.model).create_model()
    model_args, remaining = model.parse_args(remaining)
    if v2:
        # train_larvaV2.py:73-81: an "epoch" of 300 MiB of input values; volume_per_step is NOT set there, so the model's
        # stays 0 and the volume-triggered validation / checkpoint of train_step_larva never fires after step 1
        #  -- reproduced as is
        if args.steps_per_epoch is None:
            args.steps_per_epoch = round_to_1(300 * (1024 ** 2) / ((args.input_patch_size ** 2) * args.batch_size * 3))
        model.steps_per_epoch = int(args.steps_per_epoch)
    else:
        model.volume_per_step = (args.input_patch_size ** 2) * args.batch_size * 3 * world
    model.prepare(is_training=True, scales=scales, global_step=args.global_step)
    if hasattr(model, "strict_graph") and not args.allow_eager_fallback:
        model.strict_graph = True   # a failed hipGraph capture is an error here, not a silent 2.4x slowdown
    model.sync_loss = not args.async_loss
    if remaining:
        print("WARNING: found unhandled arguments: %s" % remaining)
    if args.restore_path is not None:
        model.restore(ckpt_path=args.restore_path, target=args.restore_target)
        print("restored the model")
    if args.resume_state is not None:
        model.restore_training_state(args.resume_state)
        print("restored the training state at step %d" % model.global_step)

    writers = {s: (make_summary_writer(os.path.join(args.train_path, "x%d" % s)) if rank == 0 else _NullSummary())
               for s in scales}
    if rank == 0:
        merged = {**vars(args), **(vars(loader_args) if loader_args else {}),
                  **(vars(val_args) if val_args else {}), **vars(model_args), "world_size": world}
        with open(os.path.join(args.train_path, "arguments.json"), "w") as f:
            f.write(json.dumps(merged, sort_keys=True, indent=2))

    if loader.is_threaded:
        loader.start_training_queue_runner(batch_size=args.batch_size, input_patch_size=args.input_patch_size)

    print("begin training")
    if v2:
        print(f"{model.steps_per_epoch} steps equal to 1 epoch")
    else:
        print(f"volume {model.volume_per_step/1e6:.2f}M for 1 step.")
        print(f"needs {model_args.val_volume/model.volume_per_step:.0f}steps to validate "
              f"for {model_args.val_volume/1e9:.1f}G volume.")
    log_until = model.steps_per_epoch * 2 if v2 else 1000   # (train_larvaV2.py:142 / train_larva.py:136)
    try:
        while model.global_step < args.max_steps:
            scale = model.get_next_train_scale()
            summary = writers[scale] if model.global_step % args.summary_freq == 0 else None
            t0 = time.time()
            if getattr(loader, "is_device", False):
                # dataset resident in HBM: the batch is cut, augmented and converted on the GPU
                p, b = args.input_patch_size, args.batch_size
                bufs = model.input_buffers((b, 3, p, p), (b, 3, p * scale, p * scale)) \
                    if hasattr(model, "input_buffers") else None
                input_tensor, truth_tensor = loader.get_device_batch(batch_size=b, scale=scale, input_patch_size=p,
                                                                     out=bufs)
                t1 = time.time()
            else:
                if loader.is_threaded:
                    input_list, truth_list = loader.get_queue_data(scale=scale)
                else:
                    input_list, truth_list = loader.get_patch_batch(batch_size=args.batch_size, scale=scale,
                                                                    input_patch_size=args.input_patch_size)
                t1 = time.time()
                # rot90/flip produce negative-stride views: one contiguous host array, one H2D copy
                input_tensor = torch.as_tensor(np.ascontiguousarray(np.stack(input_list), dtype=np.float32),
                                               device=model.device)
                truth_tensor = torch.as_tensor(np.ascontiguousarray(np.stack(truth_list), dtype=np.float32),
                                               device=model.device)
            t2 = time.time()
            loss = model.train_step_larva(args=args, val_dataloader=val_loader, input_tensor=input_tensor,
                                          truth_tensor=truth_tensor, summary=summary)
            t3 = time.time()
            if args.sleep_ratio > 0 and t3 > t0:
                time.sleep(min(10.0, (t3 - t0) * args.sleep_ratio))
            if rank == 0 and model.global_step < log_until and model.global_step % args.log_freq == 0:
                print("step %d, lr %.10f, loss %.6f (%.3f sec/batch)" % (model.global_step, model.get_lr(),
                                                                        float(loss), t3 - t0))
                print(f"dataload_time:{t1 - t0:.4f}s, np2ts_time:{t2 - t1:.4f}s, train_time: {t3 - t2:.4f}s")
    except KeyboardInterrupt:
        print("interrupted (KeyboardInterrupt)")

    print("finished")
    for w in writers.values():
        w.close()
    if loader.is_threaded:
        loader.stop_queue_runners()
    return model


if __name__ == "__main__":
    main()
