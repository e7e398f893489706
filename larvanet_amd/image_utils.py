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


def band_rows(height, world):
    """Rows [r0, r1) of each of `world` contiguous bands of an image `height` rows tall (balanced,
    empty bands when world > height)."""
    return [((height * r) // world, (height * (r + 1)) // world) for r in range(world)]


def upscale_band(model, input_image, scale, r0, r1, halo):
    """Output rows [scale*r0, scale*r1) of model.upscale(input_image), computed from the input rows
    [r0 - halo, r1 + halo) only.  EXACT (bit for bit) when `halo` >= the network's receptive halo
    (model.receptive_halo()): the cut edges see zero padding / clamped bicubic taps instead of the
    neighbouring rows, but that only reaches `halo` rows into the sub-image, and exactly those rows
    are dropped; true image borders stay borders.  SURVEY 8e row 3 (the reference itself only has
    the approximate 2x2 chop_forward, utils/image_utils.py:30-66)."""
    height = input_image.shape[1]
    if r1 <= r0:
        return np.zeros((input_image.shape[0], 0, input_image.shape[2] * scale), np.float32)
    lo, hi = max(0, r0 - halo), min(height, r1 + halo)
    out = model.upscale(input_list=[np.ascontiguousarray(input_image[:, lo:hi, :])], scale=scale)[0]
    return out[:, (r0 - lo) * scale:(r1 - lo) * scale, :]


def upscale_banded_device(model, input_image, scale, rank, world, all_gather, halo=None):
    """upscale_banded with the bands moved by a DEVICE collective: rank r upscales band r (+ halo)
    on its GPU, crops it on the GPU into a buffer of the tallest band's size, and
    `all_gather(tensor) -> [world][...]` (larvanet_amd.dist.all_gather_tensor = one RCCL all-gather)
    hands every rank all bands; the concatenation is a device tensor [C][scale*H][scale*W].  Nothing
    is pickled and no band visits the host (upscale_banded gathers numpy arrays as Python objects:
    a 33 MB DIV2K output would be pickled once per rank).  Bit-identical to the full-image forward
    for halo >= model.receptive_halo()."""
    import torch
    halo = model.receptive_halo() if halo is None else halo
    channels, height, width = input_image.shape
    rows = band_rows(height, world)
    tallest = max(b - a for a, b in rows)
    r0, r1 = rows[rank]
    buf = torch.zeros((channels, tallest * scale, width * scale), dtype=torch.float32, device=model.device)
    if r1 > r0:
        lo, hi = max(0, r0 - halo), min(height, r1 + halo)
        out = model.upscale_tensor(input_list=[np.ascontiguousarray(input_image[:, lo:hi, :])])[0]
        buf[:, :(r1 - r0) * scale] = out[:, (r0 - lo) * scale:(r1 - lo) * scale]
    bands = all_gather(buf)
    return torch.cat([bands[q, :, :(b - a) * scale] for q, (a, b) in enumerate(rows)], dim=1)


def upscale_banded(model, input_image, scale, rank, world, gather, halo=None):
    """One image split into `world` row bands, band r computed by rank r, `gather(obj)` = every
    rank's object in rank order (larvanet_amd.dist.gather_objects).  Every rank returns the whole
    output image.  Latency mode: each rank also recomputes 2 * halo rows, so 8 GPUs on a 339-row
    image do 113 rows each instead of 339, not 42."""
    halo = model.receptive_halo() if halo is None else halo
    r0, r1 = band_rows(input_image.shape[1], world)[rank]
    mine = upscale_band(model, input_image, scale, r0, r1, halo)
    return np.concatenate(gather(mine), axis=1)
