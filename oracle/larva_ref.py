"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.

numpy glue over oracle/larva_ref.c: every arithmetic step is the C restatement, this file only
wires the steps into the reference's network graph (citations are file:line into the reference).
Weights are passed as a dict with the reference's state_dict key names.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblarva_ref.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_u = ctypes.POINTER(ctypes.c_uint)
_b = ctypes.POINTER(ctypes.c_ubyte)


def build():
    src = os.path.join(_HERE, "larva_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i, ll, fl, d = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double
        L.ref_conv3x3.argtypes = [_f, _f, _f, _f, i, i, i, i, i]
        L.ref_conv3x3_dgrad.argtypes = [_f, _f, _f, i, i, i, i, i]
        L.ref_conv3x3_wgrad.argtypes = [_f, _f, _f, _f, i, i, i, i, i]
        L.ref_pixel_shuffle.argtypes = [_u, _u, i, i, i, i, i]
        L.ref_pixel_unshuffle.argtypes = [_u, _u, i, i, i, i, i]
        L.ref_bicubic_up.argtypes = [_f, _f, i, i, i, i]
        L.ref_l1_mean.argtypes = [_f, _f, ll]
        L.ref_l1_mean.restype = d
        L.ref_l1_grad.argtypes = [_f, _f, fl, ll, _f]
        L.ref_adamw.argtypes = [_f, _f, _f, _f, ll, i, d, d, d, d, d]
        L.ref_image_to_uint8.argtypes = [_f, _b, ll]
        L.ref_image_psnr.argtypes = [_b, i, i, i, _b, i, i]
        L.ref_image_psnr.restype = d
        _lib = L
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f):
    return a.ctypes.data_as(t)


# --- single steps ---------------------------------------------------------------------------
def conv3x3(x, w, b=None):
    x, w = _c(x), _c(w)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert w.shape == (Cout, Cin, 3, 3)
    out = np.empty((N, Cout, H, W), np.float32)
    bb = _c(b) if b is not None else None
    lib().ref_conv3x3(_p(x), _p(w), _p(bb) if bb is not None else None, _p(out), N, Cin, Cout, H, W)
    return out


def conv3x3_dgrad(dy, w):
    dy, w = _c(dy), _c(w)
    N, Cout, H, W = dy.shape
    Cin = w.shape[1]
    dx = np.empty((N, Cin, H, W), np.float32)
    lib().ref_conv3x3_dgrad(_p(dy), _p(w), _p(dx), N, Cin, Cout, H, W)
    return dx


def conv3x3_wgrad(dy, x):
    dy, x = _c(dy), _c(x)
    N, Cout, H, W = dy.shape
    Cin = x.shape[1]
    dw = np.empty((Cout, Cin, 3, 3), np.float32)
    db = np.empty((Cout,), np.float32)
    lib().ref_conv3x3_wgrad(_p(dy), _p(x), _p(dw), _p(db), N, Cin, Cout, H, W)
    return dw, db


def pixel_shuffle(x, r=4):
    x = np.ascontiguousarray(x)
    assert x.dtype.itemsize == 4
    N, C, H, W = x.shape
    out = np.empty((N, C // (r * r), H * r, W * r), x.dtype)
    lib().ref_pixel_shuffle(_p(x, _u), _p(out, _u), N, C // (r * r), H, W, r)
    return out


def pixel_unshuffle(x, r=4):
    x = np.ascontiguousarray(x)
    assert x.dtype.itemsize == 4
    N, C, HH, WW = x.shape
    out = np.empty((N, C * r * r, HH // r, WW // r), x.dtype)
    lib().ref_pixel_unshuffle(_p(x, _u), _p(out, _u), N, C, HH // r, WW // r, r)
    return out


def bicubic_up(x, scale=4):
    x = _c(x)
    N, C, H, W = x.shape
    out = np.empty((N, C, H * scale, W * scale), np.float32)
    lib().ref_bicubic_up(_p(x), _p(out), N * C, H, W, scale)
    return out


def l1_mean(a, b):
    a, b = _c(a), _c(b)
    return float(lib().ref_l1_mean(_p(a), _p(b), a.size))


def l1_grad(a, b, g=1.0):
    a, b = _c(a), _c(b)
    ga = np.empty_like(a)
    lib().ref_l1_grad(_p(a), _p(b), float(g), a.size, _p(ga))
    return ga


def adamw(p, g, m, v, step, lr=4e-4, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    p, g, m, v = (_c(t).copy() for t in (p, g, m, v))
    lib().ref_adamw(_p(p), _p(g), _p(m), _p(v), p.size, int(step), lr, beta1, beta2, eps, wd)
    return p, m, v


def image_to_uint8(img):
    """validate.py:17-18"""
    img = _c(img)
    out = np.empty(img.shape, np.uint8)
    lib().ref_image_to_uint8(_p(img), _p(out, _b), img.size)
    return out


def fit_truth_image_size(output_image, truth_image):
    """validate.py:20-21"""
    return truth_image[:, 0:output_image.shape[1], 0:output_image.shape[2]]


def image_psnr(output_image, truth_image):
    """validate.py:23-27 (truth may be larger; it is cropped top-left)."""
    o = np.ascontiguousarray(output_image, dtype=np.uint8)
    t = np.ascontiguousarray(truth_image, dtype=np.uint8)
    C, H, W = o.shape
    return float(lib().ref_image_psnr(_p(o, _b), C, H, W, _p(t, _b), t.shape[1], t.shape[2]))


# --- network graph --------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, np.float32(0))


def residual_block(sd, prefix, x):
    """models/LarvaNet.py:205-220"""
    h = relu(conv3x3(x, sd[prefix + ".body.0.weight"], sd[prefix + ".body.0.bias"]))
    return x + conv3x3(h, sd[prefix + ".body.2.weight"], sd[prefix + ".body.2.bias"])


def body(sd, i, x, num_blocks):
    """models/LarvaNet.py:236-248"""
    fea = x
    for j in range(num_blocks):
        fea = residual_block(sd, "body_%d.res_blocks.%d" % (i, j), fea)
    return x + fea


def leg(sd, prefix, fea, base):
    """models/LarvaNet.py:251-267 (prefix 'body_i.leg' or, for V2, 'tail')"""
    h = relu(conv3x3(fea, sd[prefix + ".recon_block.0.weight"], sd[prefix + ".recon_block.0.bias"]))
    out = pixel_shuffle(conv3x3(h, sd[prefix + ".recon_block.2.weight"], sd[prefix + ".recon_block.2.bias"]), 4)
    return out + base


def head(sd, x):
    """models/LarvaNet.py:223-233"""
    return conv3x3(x, sd["head.feature_extraction.weight"], sd["head.feature_extraction.bias"])


def forward_exits(sd, x, blocks):
    """All exits, as train_step_larva walks them (models/LarvaNet.py:102-108)."""
    fea = head(sd, x)
    base = bicubic_up(x, 4)
    outs, feats = [], []
    for i, nb in enumerate(blocks):
        fea = body(sd, i, fea, nb)
        feats.append(fea)
        outs.append(leg(sd, "body_%d.leg" % i, fea, base))
    return outs, feats, base


def forward(sd, x, blocks):
    """LarvaNetModule.forward, models/LarvaNet.py:287-293: last exit only."""
    return forward_exits(sd, x, blocks)[0][-1]


def forward_v2(sd, x, blocks):
    """models/LarvaNetV2.py:314-334, 355-365: cat(features) -> merge conv -> recon -> shuffle -> + base."""
    _, feats, base = forward_exits(sd, x, blocks)
    fea = conv3x3(np.concatenate(feats, axis=1), sd["tail.merge_conv.weight"], sd["tail.merge_conv.bias"])
    return leg(sd, "tail", fea, base)


def multi_exit_loss(sd, x, truth, blocks):
    """models/LarvaNet.py:104-109: sum of per-exit L1 means / num_modules (float32 like torch)."""
    outs, _, _ = forward_exits(sd, x, blocks)
    loss = np.float32(0)
    for o in outs:
        loss = np.float32(loss + np.float32(l1_mean(o, truth)))
    return float(np.float32(loss / np.float32(len(outs))))


# --- chop-forward (utils/image_utils.py:30-66) ----------------------------------------------
def split_image(image, overlap_size):
    _, height, width = image.shape
    sh, sw, ho = height // 2, width // 2, overlap_size // 2
    return [image[:, :sh + ho, :sw + ho].copy(), image[:, :sh + ho, sw - ho:].copy(),
            image[:, sh - ho:, :sw + ho].copy(), image[:, sh - ho:, sw - ho:].copy()]


def combine_images(images, input_shape, scale, overlap_size):
    _, height, width = input_shape
    sh, sw = height // 2, width // 2
    nsh, nsw, nho = sh * scale, sw * scale, (overlap_size // 2) * scale
    out = np.zeros([3, height * scale, width * scale])
    out[:, :nsh, :nsw] = images[0][:, :nsh, :nsw]
    out[:, :nsh, nsw:] = images[1][:, :nsh, nho:]
    out[:, nsh:, :nsw] = images[2][:, nho:, :nsw]
    out[:, nsh:, nsw:] = images[3][:, nho:, nho:]
    return out
