"""Builds the C-ABI HIP library (gfx950 only) in-tree: larvanet_amd/csrc/liblarva_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liblarva_hip.so")
SOURCES = ["conv3x3_mfma.hip", "wgrad3x3_mfma.hip", "larva_pointwise.hip"]
HEADERS = ["larva_common.h", "larva_bicubic.h", "larva_loss.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the larvanet_amd HIP library cannot be built")
    return exe


def _check_ring_waits(temps, verbose):
    """The conv kernel's counted waits (`s_waitcnt vmcnt(NPW + kAuxLoads)`: run_role, kAuxLate) are only correct if the
    compiler emits exactly the vector loads the constant counts between chunk 1's LDS-DMA pieces and the wait; one load
    fewer and a wait can pass with a piece still in flight -- silently wrong results after a compiler bump or a new
    instantiation.  tools/check_aux_loads.py reads the device assembly of THIS build and fails it on a mismatch."""
    import glob
    asm = glob.glob(os.path.join(temps, "conv3x3_mfma-hip-amdgcn-*.s"))
    obj = os.path.join(temps, "conv3x3_mfma.o")
    if not asm or not os.path.exists(obj):
        raise RuntimeError("larvanet_amd.build: no device assembly from the conv kernel's compile (-save-temps)")
    checker = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "tools", "check_aux_loads.py")
    r = subprocess.run([sys.executable, checker, asm[0]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = r.stdout.decode(errors="replace").strip().splitlines()[-1:]
    if verbose:
        print("[larvanet_amd.build] ring waits vs emitted loads (tools/check_aux_loads.py):", " ".join(tail), flush=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode(errors="replace"))
        raise RuntimeError("larvanet_amd.build: a counted vmcnt wait of conv3x3_mfma.hip does not match the loads hipcc emitted")
    os.replace(obj, os.path.join(CSRC, "conv3x3_mfma.o"))   # (atomic: a concurrent build never links a half-written object)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force=False, verbose=True):
    """Compile every .hip source for gfx950 and link liblarva_hip.so. Returns the library path.  The conv kernel's
    -save-temps output goes to a directory of THIS invocation (tempfile.mkdtemp under csrc/, git-ignored `build*`), so two
    builds of one checkout (several ranks, CI jobs) cannot delete or read each other's temporaries."""
    import tempfile
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    temps = None
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if src == "conv3x3_mfma.hip":   # keep the device assembly for _check_ring_waits
                temps = tempfile.mkdtemp(prefix="build_", dir=CSRC)
                cmd = [hipcc] + FLAGS + ["-save-temps=obj", "-c", s, "-o", os.path.join(temps, "conv3x3_mfma.o")]
            if verbose:
                print("[larvanet_amd.build]", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    try:
        for cmd, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out.decode(errors="replace"))
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if temps is not None:
            _check_ring_waits(temps, verbose)
    finally:
        if temps is not None:
            shutil.rmtree(temps, ignore_errors=True)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[larvanet_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_extension(force="--force" in sys.argv)
    print(LIB)
