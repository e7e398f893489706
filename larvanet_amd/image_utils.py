"""Chop-forward of the reference (utils/image_utils.py:7-66): the LR image is cut into 2x2
overlapping quadrants, each is upscaled on its own and the results are stitched at the quadrant
boundaries.  This is approximate by design (overlap/2 = 10 LR px is less than the network's
receptive-field radius of 35) and is reproduced as is."""
import numpy as np


def split_quadrants(image, overlap_size):
    _, h, w = image.shape
    sh, sw, ho = h // 2, w // 2, overlap_size // 2
    rows = (slice(None, sh + ho), slice(sh - ho, None))
    cols = (slice(None, sw + ho), slice(sw - ho, None))
    return [np.array(image[:, r, c]) for r in rows for c in cols]


def stitch_quadrants(parts, input_shape, scale, overlap_size):
    _, h, w = input_shape
    top, left = (h // 2) * scale, (w // 2) * scale
    skip = (overlap_size // 2) * scale
    out = np.zeros([3, h * scale, w * scale])
    out[:, :top, :left] = parts[0][:, :top, :left]
    out[:, :top, left:] = parts[1][:, :top, skip:]
    out[:, top:, :left] = parts[2][:, skip:, :left]
    out[:, top:, left:] = parts[3][:, skip:, skip:]
    return out


def upscale_with_chop_forward(model, input_image, scale, overlap_size):
    parts = [model.upscale(input_list=[q], scale=scale)[0] for q in split_quadrants(input_image, overlap_size)]
    return stitch_quadrants(parts, input_image.shape, scale, overlap_size)
