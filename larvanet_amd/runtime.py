"""Latency probe: counterpart of the reference's runtime.py (:57-73): per validation image one
forward pass bracketed by device synchronisations, host->device copy excluded.

    python -m larvanet_amd.runtime --model=LarvaNet --num_modules=4 --num_blocks=4,4,4,4 \\
        --dataloader=div2k_val_loader --val_input_path=... --val_truth_path=... [--restore_path=...]
"""
import argparse
import importlib
import os
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--dataloader", type=str, default="div2k_val_loader")
    p.add_argument("--model", type=str, default="LarvaNet")
    p.add_argument("--scales", type=str, default="4")
    p.add_argument("--cuda_device", type=str, default=None)
    p.add_argument("--restore_path", type=str, default=None)
    p.add_argument("--allow_eager_fallback", action="store_true",
                   help="if a hipGraph capture fails, go on with one launch per kernel (~2.4x slower) instead of raising")
    p.add_argument("--repeats", type=int, default=1, help="timed forwards per image (after one warm-up)")
    return p


def main(argv=None):
    args, remaining = build_parser().parse_known_args(argv)
    if args.cuda_device is not None:
        os.environ["HIP_VISIBLE_DEVICES"] = args.cuda_device
    scales = [int(s) for s in args.scales.split(",")]
    print("prepare data loader - %s" % args.dataloader)
    loader = importlib.import_module("larvanet_amd.dataloaders." + args.dataloader).create_loader()
    _, remaining = loader.parse_args(remaining)
    loader.prepare(scales=scales)
    print("prepare model - %s" % args.model)
    model = importlib.import_module("larvanet_amd.models." + args.model).create_model()
    _, remaining = model.parse_args(remaining)
    model.prepare(is_training=False, scales=scales)
    if hasattr(model, "strict_graph") and not args.allow_eager_fallback:
        model.strict_graph = True   # a failed hipGraph capture is an error here, not a silent 2.4x slowdown
    if remaining:
        print("WARNING: found unhandled arguments: %s" % remaining)
    if args.restore_path is not None:
        model.restore(ckpt_path=args.restore_path)
    print("begin runtime check")
    results = {}
    for scale in scales:
        runtimes = []
        with torch.no_grad():
            for index in range(loader.get_num_images()):
                lr, _, _ = loader.get_image_pair(image_index=index, scale=scale)
                x = torch.from_numpy(np.ascontiguousarray(np.asarray(lr, dtype=np.float32)[None])).to(model.device)
                model.fwd_runtime(input_tensor=x)  # warm-up: kernel attributes, weight packing
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.repeats):
                    model.fwd_runtime(input_tensor=x)
                torch.cuda.synchronize()
                rt = (time.perf_counter() - t0) / args.repeats
                runtimes.append(rt)
                print(f"{index+1}/{loader.get_num_images()}, runtime={rt:.4f}")
        results[scale] = float(np.mean(runtimes)) if runtimes else float("nan")
        print(f"runtime={results[scale]:.4f}")
    print("finished")
    return results


if __name__ == "__main__":
    main()
