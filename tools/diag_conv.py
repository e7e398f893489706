#!/usr/bin/env python3
"""Where does the fused conv kernel spend its time?  Builds timing-only ablations of
conv3x3_mfma.hip (-DLARVA_DIAG=mask, see the top of that file) and times each at the BASELINE
layer shape (16x48x48x48) with event pairs.  Outputs of the ablated builds are wrong by
construction; only the durations mean anything.

  python tools/diag_conv.py --build     (build container: hipcc cross-compiles the variants)
  python tools/diag_conv.py             (GPU box: time them)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "larvanet_amd", "csrc", "conv3x3_mfma.hip")
VARIANTS = {0: "full kernel", 1: "no MFMA", 2: "no staging loads", 4: "no epilogue traffic", 3: "no MFMA, no staging",
            6: "MFMA only", 7: "roles + barriers only", 8: "empty launch", 16: "plain output stores"}


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for v in VARIANTS:
        so = os.path.join(OUT, "libconv_diag%d.so" % v)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLARVA_DIAG=%d" % v,
               "-DLARVA_DIAG_ONLY48=1", SRC, "-o", so]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0


def main():
    import numpy as np
    import torch
    from larvanet_amd import hip_lib, kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(16, 48, 48, 48, generator=g) * 20).to(dev)
    r0 = (torch.randn(16, 48, 48, 48, generator=g) * 20).to(dev)
    r1 = (torch.randn(16, 48, 48, 48, generator=g) * 20).to(dev)
    base = (torch.randn(16, 3, 192, 192, generator=g) * 20).to(dev)
    w = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(48, device=dev)
    fwd, _ = K.pack_weights(w)
    out = torch.empty_like(x)
    out_hr = torch.empty_like(base)
    import ctypes as ct
    sig = hip_lib.SIGNATURES["larva_conv3x3_fwd_timed"]
    stream = torch.cuda.current_stream().cuda_stream
    epis = {"relu": dict(relu=1), "res1": dict(res0=r0), "res2": dict(res0=r0, res1=r1), "mask": dict(mask=r0),
            "shuffle+base": dict(mode=1, base=base)}
    print("kernel-attached event timing (mean of 40 launches), us")
    print("%-26s" % "variant" + "".join("%14s" % e for e in epis))
    for v, label in VARIANTS.items():
        lib = ctypes.CDLL(os.path.join(OUT, "libconv_diag%d.so" % v))
        fn = lib.larva_conv3x3_fwd_timed
        fn.restype, fn.argtypes = sig
        row = []
        for e, kw in epis.items():
            o = out_hr if kw.get("mode") else out
            mean, best = ct.c_float(0), ct.c_float(0)
            for iters in (5, 40):
                code = fn(hip_lib.ptr_array([x.data_ptr()]), 1, 48, fwd.data_ptr(), b.data_ptr(),
                          kw["res0"].data_ptr() if "res0" in kw else None,
                          kw["res1"].data_ptr() if "res1" in kw else None,
                          kw["mask"].data_ptr() if "mask" in kw else None,
                          kw["base"].data_ptr() if "base" in kw else None, o.data_ptr(), 16, 48, 48, 48,
                          kw.get("relu", 0), kw.get("mode", 0), stream, iters, ct.byref(mean), ct.byref(best))
                assert code == 0, code
            row.append(mean.value * 1e3)
        print("%-26s" % ("%d %s" % (v, label)) + "".join("%11.1f us" % t for t in row))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main()
