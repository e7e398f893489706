"""Device-resident training loader: the whole dataset is decoded ONCE, kept as uint8
in HBM (DIV2K: 800 HR + LR images ~ 5 GB of 288 GB), and every batch is cropped / rotated / flipped
/ converted by one HIP kernel per resolution -- no per-step PNG decode, numpy work or H2D copy.
Same sampling distribution and draw order as dataloaders/div2k_train_loader.py:76-98 of the
reference (image, x, y, rot90 k in 1..4, flip p = .5) from a per-rank RandomState; the draws are
made on the host (80 integers per batch) and shipped as one small tensor.

Source of the images: any loader plugin with get_image_pair() (--device_source, default
div2k_train_loader; `synthetic_loader` for tests and benchmarks).

Data parallel start-up is O(1) in the world size: RANK 0 decodes the dataset once (a thread pool over the source's
get_image_pair: PIL's PNG decoder releases the GIL) and the uint8 image tables + offset / size tables are BROADCAST to the
other ranks' HBM (dist.broadcast_tensor: RCCL over xGMI, ~5 GB once) -- eight ranks of one host do not decode 800 PNG
pairs eight times.  Every rank keeps its own draw stream (seed + 1000 rank)."""
import argparse
import copy
import importlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import dist as ldist
from .. import kernels as K
from .base import BaseLoader


def create_loader():
    return DevicePatchLoader()


def draw_batch(rng, shapes, batch_size, input_patch_size):
    """[batch][5] int32 {image, x, y, k, flip}, drawn in the reference's order."""
    draws = np.empty((batch_size, 5), np.int32)
    for b in range(batch_size):
        img = rng.randint(len(shapes))
        h, w = shapes[img]
        x = rng.randint(w - input_patch_size)
        y = rng.randint(h - input_patch_size)
        k = rng.randint(4) + 1
        flip = 1 if rng.uniform() < 0.5 else 0
        draws[b] = (img, x, y, k, flip)
    return draws


def apply_draw_numpy(draw, lr, hr, scale, p):
    """Host restatement of what the kernel does for one draw (used by the tests)."""
    _, x, y, k, flip = (int(v) for v in draw)
    a = np.rot90(lr[:, y:y + p, x:x + p], k=k, axes=(1, 2))
    b = np.rot90(hr[:, y * scale:(y + p) * scale, x * scale:(x + p) * scale], k=k, axes=(1, 2))
    if flip:
        a, b = a[:, :, ::-1], b[:, :, ::-1]
    return np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)


def decode_workers():
    """Threads of the one-time decode: this process's share of the host (dist.host_threads), at most 32."""
    return max(1, min(32, int(os.environ.get("LARVA_DECODE_THREADS", ldist.host_threads()))))


def build_host_tables(source, scales, workers=1):
    """Decode every image pair of `source` once -> ({scale: {"lr", "hr": uint8 [bytes], "lr_off", "hr_off": int64 [n],
    "lr_hw", "hr_hw": int32 [2n]}}, [(lr_h, lr_w)] of the first scale).  Images are rounded / clipped to uint8 exactly as
    the sampler's kernel expects them (PNG sources are uint8-valued already).  Host only: runs on rank 0."""
    n = source.get_num_images()
    shapes, tables = [], {}
    for scale in scales:
        def one(i, scale=scale):
            lr, hr, _ = source.get_image_pair(i, scale)
            return (np.ascontiguousarray(np.clip(np.round(lr), 0, 255).astype(np.uint8)),
                    np.ascontiguousarray(np.clip(np.round(hr), 0, 255).astype(np.uint8)))
        if workers > 1 and n > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                pairs = list(pool.map(one, range(n)))      # (map keeps the image order)
        else:
            pairs = [one(i) for i in range(n)]
        lr_off = np.zeros(n, np.int64)
        hr_off = np.zeros(n, np.int64)
        lr_off[1:] = np.cumsum([p[0].size for p in pairs])[:-1]
        hr_off[1:] = np.cumsum([p[1].size for p in pairs])[:-1]
        tables[scale] = {
            "lr": np.concatenate([p[0].ravel() for p in pairs]), "hr": np.concatenate([p[1].ravel() for p in pairs]),
            "lr_off": lr_off, "hr_off": hr_off,
            "lr_hw": np.asarray([d for p in pairs for d in p[0].shape[1:]], np.int32),
            "hr_hw": np.asarray([d for p in pairs for d in p[1].shape[1:]], np.int32)}
        if scale == scales[0]:
            shapes = [tuple(int(d) for d in p[0].shape[1:]) for p in pairs]
    return tables, shapes


TABLE_DTYPES = {"lr": torch.uint8, "hr": torch.uint8, "lr_off": torch.int64, "hr_off": torch.int64,
                "lr_hw": torch.int32, "hr_hw": torch.int32}


def share_tables(host_tables, shapes, scales, device):
    """Rank 0's decoded tables -> device tensors on EVERY rank.  Without a communicator: a plain upload.  With one: rank 0
    announces the table sizes (one small object broadcast), every rank allocates, and each table travels as one
    tensor broadcast from rank 0 (host_tables / shapes are None on the other ranks)."""
    if not ldist.active():
        return ({s: {k: torch.from_numpy(v).to(device) for k, v in t.items()} for s, t in host_tables.items()}, shapes)
    main = ldist.is_main()
    meta = ldist.broadcast_object(({s: {k: int(v.size) for k, v in host_tables[s].items()} for s in scales}, shapes) if main else None)
    sizes, shapes = meta
    out = {}
    for s in scales:
        out[s] = {}
        for k in ("lr", "hr", "lr_off", "hr_off", "lr_hw", "hr_hw"):
            t = (torch.from_numpy(host_tables[s][k]).to(device) if main
                 else torch.empty(sizes[s][k], dtype=TABLE_DTYPES[k], device=device))
            out[s][k] = ldist.broadcast_tensor(t, src=0)
    return out, [tuple(hw) for hw in shapes]


class DevicePatchLoader(BaseLoader):
    is_device = True

    def parse_args(self, args):
        parser = argparse.ArgumentParser()
        parser.add_argument("--device_source", type=str, default="div2k_train_loader",
                            help="loader plugin that supplies the images to make resident")
        parser.add_argument("--data_seed", type=int, default=None)
        self.args, remaining = parser.parse_known_args(args=args)
        self.source = importlib.import_module("larvanet_amd.dataloaders." + self.args.device_source).create_loader()
        src_args, remaining = self.source.parse_args(remaining)
        merged = copy.deepcopy(self.args)
        for k_, v_ in vars(src_args or argparse.Namespace()).items():
            setattr(merged, k_, v_)
        return merged, remaining

    def prepare(self, scales):
        if not torch.cuda.is_available():
            raise RuntimeError("larvanet_amd: device_patch_loader needs a HIP device")
        self.scale_list = scales
        self.source.prepare(scales)
        self.device = torch.device("cuda", torch.cuda.current_device())
        seed = self.args.data_seed
        self.rng = np.random.RandomState(None if seed is None else ldist.seed_for_rank(seed))
        # decode once (rank 0, thread pool), broadcast to the other ranks' HBM
        host, shapes = (build_host_tables(self.source, scales, decode_workers())
                        if (ldist.is_main() or not ldist.active()) else (None, None))
        self.tables, self.shapes = share_tables(host, shapes, scales, self.device)
        del host
        n = len(self.shapes)
        print("data: %d image pairs resident on %s (%.1f MB)" % (
            n, self.device, sum(t["lr"].numel() + t["hr"].numel() for t in self.tables.values()) / 1e6))

    def get_num_images(self):
        return len(self.shapes)

    def get_image_pair(self, image_index, scale):
        return self.source.get_image_pair(image_index, scale)

    def get_device_batch(self, batch_size, scale, input_patch_size, draws=None, out=None):
        """-> (input tensor [B][3][p][p], truth tensor [B][3][p*scale][p*scale]) on the device;
        out = (input buffer, truth buffer) to fill in place (the plugin's `input_buffers()`: the
        captured training step then reads the batch where the sampler wrote it, no copy)."""
        if draws is None:
            draws = draw_batch(self.rng, self.shapes, batch_size, input_patch_size)
        t = self.tables[scale]
        d = torch.from_numpy(np.ascontiguousarray(draws, dtype=np.int32)).to(self.device, non_blocking=True)
        ox, oy = out if out is not None else (None, None)
        x = K.gather_patches(t["lr"], t["lr_off"], t["lr_hw"], d, batch_size, input_patch_size, 1, out=ox)
        y = K.gather_patches(t["hr"], t["hr_off"], t["hr_hw"], d, batch_size, input_patch_size * scale, scale, out=oy)
        return x, y

    def get_patch_batch(self, batch_size, scale, input_patch_size):
        x, y = self.get_device_batch(batch_size, scale, input_patch_size)
        return list(x.cpu().numpy()), list(y.cpu().numpy())
