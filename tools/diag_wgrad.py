#!/usr/bin/env python3
"""Where does the pipelined weight-gradient kernel spend its time?  Builds timing-only ablations
of wgrad3x3_mfma.hip (-DWG_DIAG=mask, see the top of that file) and times the partial-image
launch alone at 32 layers x 8 workgroups (32 tiles per workgroup, 16x48x48x48 operands).

  python tools/diag_wgrad.py --build     (build container)
  python tools/diag_wgrad.py             (GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "larvanet_amd", "csrc", "wgrad3x3_mfma.hip")
LAGS = {}   # (the load -> LDS-write distance was a build switch until round 5: 3 / 10 / 15 k-steps measured no better than 6)
VARIANTS = {0: "full kernel", 1: "no MFMA", 2: "no staging", 4: "no operand reads", 8: "no bias sums",
            16: "no LDS writes", 32: "no address math", 64: "no global loads", 96: "no addr, no loads",
            144: "loads waited, no write", 3: "noMFMA no staging", 5: "noMFMA no reads", 17: "noMFMA no LDS writes",
            33: "noMFMA no addr math", 65: "noMFMA no loads", 7: "noMFMA bias+skeleton", 6: "MFMA only (+bias)", 14: "MFMA only", 15: "loop skeleton"}


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    only = [int(a) for a in sys.argv[2:]]
    for v in list(VARIANTS) + list(LAGS):
        if only and v not in only:
            continue
        so = os.path.join(OUT, "libwgrad_diag%d.so" % v)
        flags = ["-DWG_DIAG=%d" % v] if v in VARIANTS else ["-DWG_DIAG=%d" % LAGS[v][1], "-DWG_LAG=%d" % LAGS[v][2]]
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + flags + [
               "-I", os.path.dirname(SRC), SRC, "-o", so]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0


def main():
    import torch
    jobs, splits, iters, C = 32, 8, 10, 48
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    dys = [(torch.randn(16, C, 48, 48, generator=g) * 1e-3).to(dev) for _ in range(jobs)]
    xs = [(torch.randn(16, C, 48, 48, generator=g) * 20).to(dev) for _ in range(jobs)]
    pp = ctypes.c_void_p * jobs
    print("%-22s %10s %12s   (the first variant timed also warms the GPU up: run it twice)" % ("variant", "us/launch", "us/tile/WG"))
    only = [int(a) for a in sys.argv[1:]]
    names = dict(VARIANTS)
    names.update({k: v[0] for k, v in LAGS.items()})
    for v, name in names.items():
        if only and v not in only:
            continue
        lib = ctypes.CDLL(os.path.join(OUT, "libwgrad_diag%d.so" % v))
        lib.larva_wgrad_partial_floats.restype = ctypes.c_longlong
        nfl = lib.larva_wgrad_partial_floats(C, C, splits)
        parts = [torch.empty(nfl, device=dev) for _ in range(jobs)]
        used = ctypes.c_int(0)
        args = (pp(*[t.data_ptr() for t in dys]), pp(*[t.data_ptr() for t in xs]), pp(*[t.data_ptr() for t in parts]),
                jobs, splits, 16, C, C, 48, 48, ctypes.byref(used), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert lib.larva_conv3x3_wgrad_partial(*args) == 0
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                lib.larva_conv3x3_wgrad_partial(*args)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 1e3 / iters)
        print("%-22s %10.1f %12.2f" % (name, best, best / (256 * 16 / splits / 16)))


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
