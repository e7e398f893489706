"""Device-resident training loader: the whole dataset is decoded ONCE, kept as uint8
in HBM (DIV2K: 800 HR + LR images ~ 5 GB of 288 GB), and every batch is cropped / rotated / flipped
/ converted by one HIP kernel per resolution -- no per-step PNG decode, numpy work or H2D copy.
Same sampling distribution and draw order as dataloaders/div2k_train_loader.py:76-98 of the
reference (image, x, y, rot90 k in 1..4, flip p = .5) from a per-rank RandomState; the draws are
made on the host (80 integers per batch) and shipped as one small tensor.

Source of the images: any loader plugin with get_image_pair() (--device_source, default
div2k_train_loader; `synthetic_loader` for tests and benchmarks)."""
import argparse
import copy
import importlib

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
        self.shapes, self.tables = [], {}
        seed = self.args.data_seed
        self.rng = np.random.RandomState(None if seed is None else ldist.seed_for_rank(seed))
        n = self.source.get_num_images()
        for scale in scales:
            lr_chunks, hr_chunks, lr_off, hr_off, lr_hw, hr_hw = [], [], [], [], [], []
            lo = ho = 0
            for i in range(n):
                lr, hr, _ = self.source.get_image_pair(i, scale)
                lr8 = np.ascontiguousarray(np.clip(np.round(lr), 0, 255).astype(np.uint8))
                hr8 = np.ascontiguousarray(np.clip(np.round(hr), 0, 255).astype(np.uint8))
                lr_chunks.append(lr8.ravel()); hr_chunks.append(hr8.ravel())
                lr_off.append(lo); hr_off.append(ho)
                lo += lr8.size; ho += hr8.size
                lr_hw += [lr8.shape[1], lr8.shape[2]]; hr_hw += [hr8.shape[1], hr8.shape[2]]
                if scale == scales[0]:
                    self.shapes.append((lr8.shape[1], lr8.shape[2]))

            def dev(a, dt):
                return torch.from_numpy(np.asarray(a, dtype=dt)).to(self.device)

            self.tables[scale] = {
                "lr": dev(np.concatenate(lr_chunks), np.uint8), "hr": dev(np.concatenate(hr_chunks), np.uint8),
                "lr_off": dev(lr_off, np.int64), "hr_off": dev(hr_off, np.int64),
                "lr_hw": dev(lr_hw, np.int32), "hr_hw": dev(hr_hw, np.int32)}
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
