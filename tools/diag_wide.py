#!/usr/bin/env python3
"""One full-image layer of the inference path under in-kernel stamps: a 48 -> 48 conv3x3 on a
1 x 48 x 339 x 510 activation (pitch 512), the whole-tensor 3 x 48-tile launch of 1243 workgroups that validate.py's
forward issues 34 times per image (models/LarvaNet.py:283-293 of the reference), from tools/_diag/libconv_diag32.so (a
-DLARVA_DIAG=32 build: wave 0 of every workgroup stores the 100 MHz clock at entry / chunk 0 and 1 issued / first chunk
landed / K loop done / stores issued / drained).  Prints the launch's span, the workgroups' lifetimes and K-loop
shares, how many workgroups are resident over time and the tail of the grid.   usage: diag_wide.py [relu|res1|res2]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from larvanet_amd import hip_lib, kernels as K

kind = sys.argv[1] if len(sys.argv) > 1 else "relu"
dev = torch.device("cuda", 0)
H, W, P = 339, 510, 512
g = torch.Generator().manual_seed(0)
x = torch.zeros(1, 48, H, P, device=dev)
x[..., :W] = (torch.randn(1, 48, H, W, generator=g) * 20).to(dev)
r0, r1 = x.flip(1).contiguous(), x.flip(2).contiguous()
w = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
b = torch.zeros(48, device=dev)
fwd, _ = K.pack_weights(w)
out = torch.empty_like(x)
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_diag", "libconv_diag32.so"))
fn = lib.larva_conv3x3_fwd_pitched
fn.restype, fn.argtypes = hip_lib.SIGNATURES["larva_conv3x3_fwd_pitched"]
lib.larva_diag_set_stamps.argtypes = [ctypes.c_void_p]
tiles = ((H + 2) // 3) * ((P + 47) // 48)
stamps = torch.zeros(tiles * 16, device=dev, dtype=torch.int64)
stream = torch.cuda.current_stream().cuda_stream


def launch():
    code = fn(hip_lib.ptr_array([x.data_ptr()]), 1, 48, fwd.data_ptr(), b.data_ptr(),
              r0.data_ptr() if kind in ("res1", "res2") else None, r1.data_ptr() if kind == "res2" else None, None, None,
              out.data_ptr(), 1, 48, H, W, P, 1 if kind == "relu" else 0, 0, stream)
    hip_lib.check(code, "larva_conv3x3_fwd_pitched")


os.environ["LARVA_PERSIST"] = "1"
for _ in range(3):
    launch()
torch.cuda.synchronize()
assert lib.larva_diag_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
for _ in range(3):
    launch()
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(tiles, 16).astype(np.float64) / 100.0
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
end = t[:, 5] - t0
print("persistent tiles (the product's launch for this shape), conv3x3 48 -> 48 + %s on 1 x 48 x %d x %d: %d workgroups walk %d tiles; span %.1f us = "
      "%.3f of the fp32 matrix peak; workgroups enter within %.2f us; their last store drains at p10 %.1f / median %.1f / p90 %.1f / max %.1f us"
      % (kind, H, W, len(t), tiles, end.max(), 2 * 9 * 48 * 48 * H * W / (end.max() * 1e-6) / 157.3e12, (t[:, 0] - t0).max(),
         np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
late = np.sort(t[:, 0] - t0)
print("   entries: %d within 1 us, %d later (at %s us)" % (np.sum(late < 1.0), np.sum(late >= 1.0), " ".join("%.0f" % v for v in late[late >= 1.0][::16])))
stamps.zero_()
torch.cuda.synchronize()
os.environ["LARVA_PERSIST"] = "0"   # below: one workgroup per tile (the launch of rounds 1-4), every tile's own timeline
for _ in range(3):
    launch()
torch.cuda.synchronize()
assert lib.larva_diag_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
for _ in range(3):   # back to back, as the forward issues them; the LAST launch's stamps are read
    launch()
torch.cuda.synchronize()
t = stamps.cpu().numpy().reshape(tiles, 16).astype(np.float64) / 100.0
t0 = t[:, 0].min()
ent, issued, landed, kdone, stored, drained = (t[:, i] - t0 for i in (0, 1, 2, 3, 4, 5))
span = drained.max()
flop = 2 * 9 * 48 * 48 * H * W
print("one workgroup per tile (LARVA_PERSIST=0): conv3x3 48 -> 48 + %s on 1 x 48 x %d x %d (pitch %d): %d workgroups of 3 x 48 pixels, launch span %.1f us = %.3f of the fp32 matrix peak"
      % (kind, H, W, P, tiles, span, flop / (span * 1e-6) / 157.3e12))
life = drained - ent
print("workgroup life (entry -> stores drained): p10 %.1f / median %.1f / p90 %.1f us; entry -> first chunk landed %.2f | K loop %.2f | stores issued %.2f | "
      "drained %.2f (medians)" % (np.percentile(life, 10), np.median(life), np.percentile(life, 90), np.median(landed - ent),
                                  np.median(kdone - landed), np.median(stored - kdone), np.median(drained - stored)))
order = np.argsort(ent)
first_wave = np.sum(ent < 1.0)
print("entries: %d workgroups within the first microsecond (2 per CU = 512), the last one enters at %.1f us; the last to drain entered at %.1f us"
      % (first_wave, ent.max(), ent[np.argmax(drained)]))
edges = np.linspace(0, span, 21)
res = [np.mean([np.sum((ent <= s) & (drained > s)) for s in np.linspace(a, z, 9)[:-1]]) for a, z in zip(edges[:-1], edges[1:])]
ink = [np.mean([np.sum((landed <= s) & (kdone > s)) for s in np.linspace(a, z, 9)[:-1]]) for a, z in zip(edges[:-1], edges[1:])]
print("resident workgroups (of 512 slots) per twentieth of the span: " + " ".join("%d" % round(v) for v in res))
print("  ... of which inside their K loop:                           " + " ".join("%d" % round(v) for v in ink))
tail = span - np.percentile(drained, 50)
print("half of the workgroups have drained by %.1f us; the last %.0f %% of the span run with < 256 workgroups in a K loop"
      % (np.percentile(drained, 50), 100.0 * np.mean(np.array(ink) < 256)))
