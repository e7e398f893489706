"""ctypes binding of the MEASUREMENT library (tools/build_diag.sh -> tools/_diag/diag.so: the product's sources built with
-DLARVA_DIAG_API; entry points declared in tools/larva_diag.h).  bench.py's `launch_alone_ms` fields and the tools under
tools/ use it; the product package never does.  LARVA_DIAG_LIB names another build."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from larvanet_amd import hip_lib   # noqa: E402

DIAG_PATH = os.environ.get("LARVA_DIAG_LIB") or os.path.join(ROOT, "tools", "_diag", "diag.so")
_p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_float)
_pp = ctypes.POINTER(ctypes.c_void_p)
SIGNATURES = {
    "larva_conv3x3_fwd_timed": (_i, [_pp, _i, _i] + [_p] * 7 + [_i] * 6 + [_p, _i, _f, _f]),
    "larva_conv3x3_fwd_strips_timed": (_i, [_pp, _i, _i] + [_p] * 7 + [_i] * 7 + [_p, ctypes.POINTER(ctypes.c_uint), _i, _i, _p, _i, _f, _f]),
    "larva_stamp_clock": (_i, [_p, _p]),
    "larva_delay_ticks": (_i, [_i, _p]),
    "larva_clock_probe": (_i, [_i, _p, _p]),
    "larva_conv3x3_pair_chain_probe": (_i, [_p] * 4 + [_i] * 4 + [_p] * 4 + [_i, _p, _p, _p] + [_i] * 5 + [_p]),
    "larva_conv3x3_pipeline_workspace_bytes": (ctypes.c_longlong, [_i, _i, _i]),
    "larva_conv3x3_pipeline_plan_bytes": (ctypes.c_longlong, []),
    "larva_conv3x3_pipeline_plan": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "larva_conv3x3_pipeline_run": (_i, [_p, _i, _p]),
}
_lib = None


def available():
    return os.path.exists(DIAG_PATH)


def load():
    """The measurement library (RuntimeError if it has not been built).  Loaded beside liblarva_hip.so: its copy of the
    product kernels is only ever used by the measurement entry points themselves."""
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("%s is missing: tools/build_diag.sh diag" % DIAG_PATH)
        import torch  # noqa: F401  (torch owns the process's HIP runtime)
        lib = ctypes.CDLL(DIAG_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def conv3x3_relu_timed(x, wpk, cout, bias, out, iters):
    """(mean_ms, min_ms) of the fused conv + ReLU launch from kernel-attached events: the kernel's own begin / end."""
    import torch
    lib = load()
    N, cin, H, W = (int(v) for v in x.shape)
    mean, best = ctypes.c_float(0), ctypes.c_float(0)
    code = lib.larva_conv3x3_fwd_timed(hip_lib.ptr_array([x.data_ptr()]), 1, cin, wpk.data_ptr(), bias.data_ptr(), None, None, None, None,
                                       out.data_ptr(), N, cout, H, W, 1, 0, torch.cuda.current_stream().cuda_stream, iters,
                                       ctypes.byref(mean), ctypes.byref(best))
    hip_lib.check(code, "larva_conv3x3_fwd_timed")
    return float(mean.value), float(best.value)


def conv3x3_strips_timed(x, wpk, cout, bias, out, iters, images=None, phase=0, relu=False, mask=None, res0=None, res1=None,
                         plain_stores=False):
    """(mean_ms, min_ms) of ONE strip-tile launch over images [lo, hi) running alone, with the epilogue the operands select:
    the figure a profiler reports per dispatch of conv3x3_mfma_strip_kernel<cout, EPI>."""
    import torch
    from larvanet_amd import kernels as K
    lib = load()
    N, cin, H, W = (int(v) for v in x.shape)
    lo, hi = (0, N) if images is None else images
    tab = K.strip_tile_table(H, W, out.device, phase=phase)
    if tab is None:
        raise RuntimeError("no strip tiling for %d x %d" % (H, W))

    def at(t):
        return None if t is None else t.data_ptr() + 4 * lo * cout * H * W

    mean, best = ctypes.c_float(0), ctypes.c_float(0)
    code = lib.larva_conv3x3_fwd_strips_timed(
        hip_lib.ptr_array([x.data_ptr() + 4 * lo * cin * H * W]), 1, cin, wpk.data_ptr(), bias.data_ptr() if bias is not None else None,
        at(res0), at(res1), at(mask), None, out.data_ptr() + 4 * lo * cout * H * W, hi - lo, cout, H, W, W, 1 if relu else 0, 0,
        tab[0].data_ptr(), tab[2], tab[1], 1 if plain_stores else 0, torch.cuda.current_stream().cuda_stream, iters,
        ctypes.byref(mean), ctypes.byref(best))
    hip_lib.check(code, "larva_conv3x3_fwd_strips_timed")
    return float(mean.value), float(best.value)


class PipeLayer(ctypes.Structure):
    """larva_pipe_layer of tools/larva_diag.h"""
    _fields_ = [("src", ctypes.c_void_p * 8), ("n_src", ctypes.c_int), ("cin_per_src", ctypes.c_int),
                ("wpk", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("res0", ctypes.c_void_p), ("res1", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("relu", ctypes.c_int), ("dep", ctypes.c_int)]


class ConvPipeline:
    """The round-5 layer-pipeline experiment: a chain of 48 -> 48 conv3x3 layers over one [N][48][H][P] shape in ONE launch
    of persistent workgroups (larva_conv3x3_pipeline_plan / _run; csrc/conv3x3_pipe.inc).  layers: dicts {srcs: [tensor, ...],
    wpk, bias, relu, res0, res1, out, dep}; every `out` is a tensor of its own, dep = index of the layer that writes srcs
    (-1: written before the launch).  The object keeps every tensor referenced; run() is stream-ordered and capturable;
    check() raises when a wait inside a launch expired (its outputs are garbage) -- exact after a synchronisation."""

    def __init__(self, layers, logical_w=None):
        import torch
        from larvanet_amd import kernels as K
        lib = load()
        if not layers:
            raise RuntimeError("an empty layer pipeline")
        first = layers[0]["srcs"][0]
        N, _, H, P = (int(v) for v in first.shape)
        W = P if logical_w is None else int(logical_w)
        full = (N, 48, H, P)
        arr = (PipeLayer * len(layers))()
        self._keep = []
        for i, l in enumerate(layers):
            srcs = list(l["srcs"])
            cps = int(srcs[0].shape[1])
            for k, t in enumerate(srcs):
                arr[i].src[k] = K._chk(t, "layer %d src[%d]" % (i, k), (N, cps, H, P))
            arr[i].n_src, arr[i].cin_per_src = len(srcs), cps
            K._chk(l["wpk"], "layer %d wpk" % i, (K.packed_weight_floats(48, cps * len(srcs)),))
            arr[i].wpk = l["wpk"].data_ptr()
            arr[i].bias = K._opt(l.get("bias"), "bias", (48,))
            arr[i].res0 = K._opt(l.get("res0"), "res0", full)
            arr[i].res1 = K._opt(l.get("res1"), "res1", full)
            arr[i].out = K._chk(l["out"], "layer %d out" % i, full)
            arr[i].relu, arr[i].dep = (1 if l.get("relu") else 0), int(l["dep"])
            self._keep += srcs + [l["wpk"], l.get("bias"), l.get("res0"), l.get("res1"), l["out"]]
        dev = first.device
        nbytes = int(lib.larva_conv3x3_pipeline_workspace_bytes(len(layers), N, H))
        if nbytes <= 0:
            raise RuntimeError("unsupported pipeline size (%d layers)" % len(layers))
        self.workspace = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
        self.error = torch.zeros(1, dtype=torch.int32).pin_memory()   # the kernel writes it, the host reads it
        self.plan = ctypes.create_string_buffer(int(lib.larva_conv3x3_pipeline_plan_bytes()))
        torch.cuda.current_stream(dev).synchronize()   # (the table copy is synchronous and the zero fill above is not)
        code = lib.larva_conv3x3_pipeline_plan(ctypes.addressof(arr), len(layers), N, 48, H, W, P, self.workspace.data_ptr(),
                                               self.error.data_ptr(), ctypes.addressof(self.plan))
        self.supported = code != 801   # hipErrorNotSupported: unaligned operands
        if self.supported:
            hip_lib.check(code, "larva_conv3x3_pipeline_plan")
        self.layers = len(layers)

    def run(self, spin_limit=0):
        import torch
        hip_lib.check(load().larva_conv3x3_pipeline_run(ctypes.addressof(self.plan), int(spin_limit), torch.cuda.current_stream().cuda_stream),
                      "larva_conv3x3_pipeline_run")

    def check(self):
        if int(self.error[0]) != 0:
            raise RuntimeError("a wait inside the layer-pipeline launch expired; its outputs are invalid")
