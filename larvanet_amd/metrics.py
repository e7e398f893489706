"""Validation metric of the reference (validate.py:17-27), host side, numpy like the reference:
uint8 conversion by round-half-to-even + clip, top-left crop of the truth to the output size,
PSNR over all RGB pixels (no border shave, no Y conversion)."""
import numpy as np


def image_to_uint8(image):
    return np.clip(np.round(image), a_min=0, a_max=255).astype(np.uint8)


def fit_truth_image_size(output_image, truth_image):
    return truth_image[:, 0:output_image.shape[1], 0:output_image.shape[2]]


def image_psnr(output_image, truth_image):
    diff = np.float32(truth_image) - np.float32(output_image)
    mse = np.mean(np.power(diff, 2))
    return 10.0 * np.log10(255.0 ** 2 / mse)
