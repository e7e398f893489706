"""DIV2K random-patch sampler (dataloaders/div2k_train_loader.py:50-148 of the reference):
random image -> random LR crop + the aligned HR crop -> rot90 by k in {1..4} -> horizontal flip
with p = 0.5.  PNGs are read with PIL (lossless, RGB values identical to the reference's
cv2.imread + BGR2RGB); the RNG is a per-loader RandomState so that data-parallel ranks draw
different patches (the reference uses the un-seeded global numpy RNG)."""
import argparse
import copy
import os

import numpy as np

from .. import dist as ldist
from .base import BaseLoader


def create_loader():
    return DIV2KLoader()


def load_png_chw(path, as_float):
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    arr = np.transpose(arr, [2, 0, 1])
    return arr.astype(np.float32) if as_float else np.ascontiguousarray(arr)


def augment_pair(rng, input_image, truth_image, scale, input_patch_size):
    """Crop / rotate / flip exactly as dataloaders/div2k_train_loader.py:76-98, drawing from `rng`
    in the same order (x, y, rot, flip)."""
    _, height, width = input_image.shape
    x = rng.randint(width - input_patch_size)
    y = rng.randint(height - input_patch_size)
    p, tp = input_patch_size, input_patch_size * scale
    lr = input_image[:, y:y + p, x:x + p]
    hr = truth_image[:, y * scale:y * scale + tp, x * scale:x * scale + tp]
    k = rng.randint(4) + 1
    lr = np.rot90(lr, k=k, axes=(1, 2))
    hr = np.rot90(hr, k=k, axes=(1, 2))
    if rng.uniform() < 0.5:
        lr, hr = lr[:, :, ::-1], hr[:, :, ::-1]
    return lr, hr


class DIV2KLoader(BaseLoader):
    float_images = True

    def _add_args(self, parser):
        parser.add_argument("--data_input_path", type=str, default="data/DIV2K_train_LR_bicubic",
                            help="LR root; scale-4 images are expected in <root>/X4/<name>x4.png")
        parser.add_argument("--data_truth_path", type=str, default="data/DIV2K_train_HR")
        parser.add_argument("--data_cached", action="store_true", help="keep decoded images in host memory")
        parser.add_argument("--data_seed", type=int, default=None,
                            help="base seed of the patch sampler (rank r uses seed + 1000 r); default: entropy")

    def parse_args(self, args):
        parser = argparse.ArgumentParser()
        self._add_args(parser)
        self.args, remaining = parser.parse_known_args(args=args)
        return copy.deepcopy(self.args), remaining

    def prepare(self, scales):
        self.scale_list = scales
        names = [os.path.splitext(f)[0] for f in os.listdir(self.args.data_truth_path) if f.lower().endswith(".png")]
        self.image_name_list = sorted(names)
        print("data: %d images are prepared (%s)" % (len(names), "caching enabled" if self.args.data_cached
                                                      else "caching disabled"))
        self._cache = {}
        seed = getattr(self.args, "data_seed", None)
        self.rng = np.random.RandomState(None if seed is None else ldist.seed_for_rank(seed))

    def get_num_images(self):
        return len(self.image_name_list)

    def _image(self, key, path):
        if self.args.data_cached and key in self._cache:
            return self._cache[key]
        img = load_png_chw(path, self.float_images)
        if self.args.data_cached:
            self._cache[key] = img
        return img

    def get_image_pair(self, image_index, scale):
        name = self.image_name_list[image_index]
        lr = self._image(("lr", scale, name), os.path.join(self.args.data_input_path, "X%d" % scale,
                                                            "%sx%d.png" % (name, scale)))
        hr = self._image(("hr", name), os.path.join(self.args.data_truth_path, "%s.png" % name))
        return lr, hr, name

    def get_image_patch_pair(self, image_index, scale, input_patch_size):
        lr, hr, _ = self.get_image_pair(image_index=image_index, scale=scale)
        return augment_pair(self.rng, lr, hr, scale, input_patch_size)

    def get_random_image_patch_pair(self, scale, input_patch_size):
        return self.get_image_patch_pair(self.rng.randint(self.get_num_images()), scale, input_patch_size)

    def get_patch_batch(self, batch_size, scale, input_patch_size):
        pairs = [self.get_random_image_patch_pair(scale, input_patch_size) for _ in range(batch_size)]
        return [p[0] for p in pairs], [p[1] for p in pairs]
