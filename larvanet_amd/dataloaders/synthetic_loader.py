"""Dataset-free loader: smooth pseudo-images generated from a seed (no files, no network).  Used
by the benchmark, the smoke test and the driver tests; same plugin surface as the DIV2K loaders."""
import argparse
import copy

import numpy as np

from .. import dist as ldist
from .base import BaseLoader
from .div2k_train_loader import augment_pair


def create_loader():
    return SyntheticLoader()


def make_pair(index, lr_h, lr_w, scale, as_float):
    """HR = low-frequency colour field + texture; LR = box-filtered HR (uint8-valued)."""
    rng = np.random.RandomState(1234 + index)
    H, W = lr_h * scale, lr_w * scale
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    hr = np.stack([127 + 90 * np.sin(xx / (7.0 + c) + rng.rand() * 6) * np.cos(yy / (11.0 - c) + rng.rand() * 6)
                   for c in range(3)]) + rng.randn(3, H, W) * 12
    hr = np.clip(np.round(hr), 0, 255)
    lr = np.clip(np.round(hr.reshape(3, lr_h, scale, lr_w, scale).mean(axis=(2, 4))), 0, 255)
    if as_float:
        return lr.astype(np.float32), hr.astype(np.float32)
    return lr.astype(np.uint8), hr.astype(np.uint8)


class SyntheticLoader(BaseLoader):
    def parse_args(self, args):
        parser = argparse.ArgumentParser()
        parser.add_argument("--synthetic_images", type=int, default=8)
        parser.add_argument("--synthetic_lr_size", type=int, default=96)
        parser.add_argument("--synthetic_uint8", action="store_true", help="validation style: uint8 images")
        parser.add_argument("--data_seed", type=int, default=0)
        self.args, remaining = parser.parse_known_args(args=args)
        return copy.deepcopy(self.args), remaining

    def prepare(self, scales):
        if not hasattr(self, "args"):
            self.parse_args([])
        self.scale_list = scales
        self.rng = np.random.RandomState(ldist.seed_for_rank(self.args.data_seed))
        self._cache = {}

    def get_num_images(self):
        return self.args.synthetic_images

    def get_image_pair(self, image_index, scale):
        key = (image_index, scale)
        if key not in self._cache:
            s = self.args.synthetic_lr_size
            self._cache[key] = make_pair(image_index, s, s + 8 * (image_index % 3), scale,
                                         not self.args.synthetic_uint8)
        lr, hr = self._cache[key]
        return lr, hr, "synthetic_%04d" % image_index

    def get_image_patch_pair(self, image_index, scale, input_patch_size):
        lr, hr, _ = self.get_image_pair(image_index, scale)
        return augment_pair(self.rng, lr, hr, scale, input_patch_size)

    def get_random_image_patch_pair(self, scale, input_patch_size):
        return self.get_image_patch_pair(self.rng.randint(self.get_num_images()), scale, input_patch_size)

    def get_patch_batch(self, batch_size, scale, input_patch_size):
        pairs = [self.get_random_image_patch_pair(scale, input_patch_size) for _ in range(batch_size)]
        return [p[0] for p in pairs], [p[1] for p in pairs]
