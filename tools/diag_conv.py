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
SRC = os.path.join(ROOT, "tools", "csrc", "conv3x3_diag.hip")   # the product's conv3x3_mfma.hip + the timed-launch entry points
TIMELINE = 32   # full kernel + in-kernel wall-clock stamps (--timeline)
TIMELINE_MFMA = 38   # the same without staging and epilogue traffic
TIMELINE_CLK = 96    # full kernel, stamps in shader-clock cycles (s_memtime)
TIMELINE_MFMA_CLK = 102
VARIANTS = {0: "full kernel", 1: "no MFMA", 2: "no staging loads", 4: "no epilogue traffic", 3: "no MFMA, no staging",
            6: "MFMA only", 7: "roles + barriers only", 8: "empty launch", 16: "plain output stores"}


EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]      # e.g. -DLARVA_REGION_KSTEPS=2: applied to every variant
TAG = os.environ.get("DIAG_TAG", "")                          # library name suffix for such builds


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    only = [TIMELINE, TIMELINE_MFMA, TIMELINE_CLK, TIMELINE_MFMA_CLK, TIMELINE_MFMA | 128, TIMELINE_MFMA_CLK | 128] \
        if "--clock-only" in sys.argv else \
        list(VARIANTS) + [TIMELINE, TIMELINE_MFMA, TIMELINE_CLK, TIMELINE_MFMA_CLK]
    for v in only:
        so = os.path.join(OUT, "libconv_diag%d%s.so" % (v, TAG))
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLARVA_DIAG=%d" % v,
               "-DLARVA_DIAG_ONLY48=1", "-I" + os.path.join(ROOT, "larvanet_amd", "csrc"), "-I" + os.path.join(ROOT, "tools", "csrc")] + EXTRA + [SRC, "-o", so]
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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import diag_lib
    sig = diag_lib.SIGNATURES["larva_conv3x3_fwd_timed"]
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


def timeline():
    """Where inside its 18 us does a conv launch spend the time?  100 MHz wall-clock stamps of wave 0
    of every workgroup, relative to the earliest kernel-entry stamp of the launch; a chain of 6
    dependent launches (each reads the previous output), the last one is reported."""
    import numpy as np
    import torch
    import ctypes as ct
    from larvanet_amd import hip_lib, kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    bufs = [(torch.randn(16, 48, 48, 48, generator=g) * 20).to(dev) for _ in range(2)]
    w = (torch.randn(48, 48, 3, 3, generator=g) * 0.01).to(dev)
    b = torch.zeros(48, device=dev)
    fwd, _ = K.pack_weights(w)
    which = TIMELINE_MFMA if "--mfma-only" in sys.argv else TIMELINE
    if "--no-operand-reads" in sys.argv:
        which = TIMELINE_MFMA | 128
    if "--clock" in sys.argv:
        return clock(which)
    lib = ctypes.CDLL(os.path.join(OUT, "libconv_diag%d.so" % which))
    fn = lib.larva_conv3x3_fwd
    fn.restype, fn.argtypes = hip_lib.SIGNATURES["larva_conv3x3_fwd"]
    stamps = torch.zeros(256 * 16, device=dev, dtype=torch.int64)
    lib.larva_diag_set_stamps.argtypes = [ct.c_void_p]
    assert lib.larva_diag_set_stamps(stamps.data_ptr()) == 0
    stream = torch.cuda.current_stream().cuda_stream
    rows = []
    for rep in range(5):
        for i in range(6):
            src, dst = bufs[i & 1], bufs[(i + 1) & 1]
            assert fn(hip_lib.ptr_array([src.data_ptr()]), 1, 48, fwd.data_ptr(), b.data_ptr(), None, None, None, None,
                      dst.data_ptr(), 16, 48, 48, 48, 1, 0, stream) == 0
        torch.cuda.synchronize()
        t = stamps.cpu().numpy().reshape(256, 16).astype(np.float64) * 0.01  # us
        rows.append(t - t[:, 0].min())
    t = np.median(np.stack(rows), axis=0)
    names = ["kernel entry", "DMA of chunks 0,1 issued", "chunk 0 landed (1st barrier)", "K loop done",
             "stores issued", "stores drained"]
    print("stamp                              median over WGs   min     max    (us after the first workgroup's entry)")
    for k, nme in enumerate(names):
        print("%-34s %10.2f %10.2f %7.2f" % (nme, np.median(t[:, k]), t[:, k].min(), t[:, k].max()))
    print("per workgroup: entry->issue %.2f, issue->landed %.2f, K loop %.2f, store issue %.2f, drain %.2f us (medians)"
          % tuple(np.median(t[:, k + 1] - t[:, k]) for k in range(5)))
    pro = [t[:, 6] - t[:, 0], t[:, 7] - t[:, 6], t[:, 14] - t[:, 7], t[:, 15] - t[:, 14], t[:, 1] - t[:, 15]]
    print("prologue: entry->chunk_src %.2f, weight plan+issue %.2f, bias/aux loads %.2f, input plan+issue %.2f, "
          "chunk 1 issue %.2f us (medians)" % tuple(np.median(v) for v in pro))
    ends = np.concatenate([t[:, 9:14], t[:, 3:4]], axis=1) - t[:, 8:13 + 1]
    print("chunk durations (barrier exit -> next barrier exit / loop end), medians: " +
          " ".join("%.2f" % np.median(ends[:, c]) for c in range(6)))


def clock(which):
    """K-loop duration in wall-clock microseconds (s_memrealtime build) and in shader-clock cycles
    (s_memtime build) of the same launch chain -> the clock the loop ran at, and how many of those
    cycles the 756 MFMAs of a wave (x 32) account for."""
    import numpy as np
    import torch
    import ctypes as ct
    from larvanet_amd import hip_lib, kernels as K
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    bufs = [(torch.randn(16, 48, 48, 48, generator=g) * 20).to(dev) for _ in range(2)]
    w = (torch.randn(48, 48, 3, 3, generator=g) * 0.01).to(dev)
    b = torch.zeros(48, device=dev)
    fwd, _ = K.pack_weights(w)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, variant in (("us", which), ("cycles", which | 64)):
        lib = ctypes.CDLL(os.path.join(OUT, "libconv_diag%d%s.so" % (variant, TAG)))
        fn = lib.larva_conv3x3_fwd
        fn.restype, fn.argtypes = hip_lib.SIGNATURES["larva_conv3x3_fwd"]
        stamps = torch.zeros(256 * 16, device=dev, dtype=torch.int64)
        lib.larva_diag_set_stamps.argtypes = [ct.c_void_p]
        assert lib.larva_diag_set_stamps(stamps.data_ptr()) == 0
        rows = []
        for rep in range(7):
            for i in range(40):   # a long chain: the clock has settled by the last launch
                src, dst = bufs[i & 1], bufs[(i + 1) & 1]
                assert fn(hip_lib.ptr_array([src.data_ptr()]), 1, 48, fwd.data_ptr(), b.data_ptr(), None, None, None, None,
                          dst.data_ptr(), 16, 48, 48, 48, 1, 0, stream) == 0
            torch.cuda.synchronize()
            t = stamps.cpu().numpy().reshape(256, 16).astype(np.float64)
            rows.append(np.stack([t[:, 3] - t[:, 2], t[:, 5] - t[:, 0]], 1))   # K loop, workgroup lifetime
        res[name] = np.median(np.stack(rows), axis=0)
    k_us, life_us = np.median(res["us"][:, 0]) * 0.01, np.median(res["us"][:, 1]) * 0.01
    k_cy, life_cy = np.median(res["cycles"][:, 0]), np.median(res["cycles"][:, 1])
    print("K loop: %.2f us, %.0f s_memtime ticks -> %.3f ticks/us; workgroup lifetime %.2f us, %.0f ticks -> %.3f ticks/us"
          % (k_us, k_cy, k_cy / k_us, life_us, life_cy, life_cy / life_us))
    print("a wave's 756 MFMAs x 32 cycles = 24192 cycles = %.2f us at 2.4 GHz, %.2f us at 2.1 GHz, %.2f us at 1.9 GHz"
          % (24192 / 2400, 24192 / 2100, 24192 / 1900))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--timeline" in sys.argv:
        timeline()
    else:
        main()
