"""ctypes binding of liblarva_hip.so (the C ABI declared in include/larva_hip.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblarva_hip.so")

_c_float_p = ctypes.c_void_p  # device pointers travel as integers
_c_pp = ctypes.POINTER(ctypes.c_void_p)
_c_int_p = ctypes.POINTER(ctypes.c_int)

# name -> (restype, argtypes); mirrors include/larva_hip.h one to one
SIGNATURES = {
    "larva_abi_version": (ctypes.c_int, []),
    "larva_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "larva_packed_weight_floats": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int]),
    "larva_pack_weights": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_pack_weights_batch": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p,
                                                ctypes.c_int, ctypes.c_void_p]),
    "larva_step_prologue": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int,
                                           _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_fwd": (ctypes.c_int, [_c_pp, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                         _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_fwd_pitched": (ctypes.c_int, [_c_pp, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                                 _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_fwd_batch": (ctypes.c_int, [ctypes.c_int, _c_pp, ctypes.c_int, ctypes.c_int, _c_pp, _c_pp, _c_pp, _c_pp,
                                               _c_pp, _c_pp, _c_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_fwd_tiled": (ctypes.c_int, [_c_pp, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                               _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_fwd_strips": (ctypes.c_int, [_c_pp, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                                _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_head_conv3_direct": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_exit_l1_partials": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "larva_conv3x3_exit_l1_batch": (ctypes.c_int, [ctypes.c_int, _c_pp, ctypes.c_int, ctypes.c_int, _c_pp, _c_pp, _c_pp, _c_pp,
                                                   _c_pp, _c_pp, _c_pp, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_strip_tile_table": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint),
                                              ctypes.c_int]),
    "larva_wgrad_partial_floats": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "larva_conv3x3_wgrad": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, _c_pp, _c_pp, _c_int_p, _c_int_p, _c_int_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_conv3x3_wgrad_partial": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   _c_int_p, ctypes.c_void_p]),
    "larva_wgrad_flat_head_splits": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "larva_conv3x3_wgrad_partial_flat_head": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, ctypes.c_int, _c_float_p, _c_float_p,
                                                             _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                             ctypes.c_int, _c_int_p, _c_int_p, ctypes.c_void_p]),
    "larva_wgrad_flat_max_splits": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "larva_conv3x3_wgrad_partial_flat": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        _c_int_p, ctypes.c_void_p]),
    "larva_wgrad_reduce": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p,
                                          _c_int_p, ctypes.c_int, ctypes.c_void_p]),
    "larva_wgrad_reduce_with_loss": (ctypes.c_int, [_c_pp, _c_pp, _c_pp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p,
                                                    _c_int_p, ctypes.c_int, _c_pp, _c_int_p, _c_float_p, ctypes.c_int,
                                                    ctypes.c_float, _c_float_p, ctypes.c_void_p]),
    "larva_adamw_step_host_copy": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_double, ctypes.c_float, ctypes.c_longlong, _c_float_p, _c_float_p,
                                                  ctypes.c_void_p]),
    "larva_upsample4_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_bicubic4_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p]),
    "larva_l1_workspace_floats": (ctypes.c_int, []),
    "larva_l1_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, _c_float_p, _c_float_p,
                                    ctypes.c_void_p]),
    "larva_l1_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_longlong, _c_float_p,
                                    ctypes.c_void_p]),
    "larva_l1_bwd_unshuffle4": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_float, _c_float_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_l1_partial": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, _c_float_p, _c_int_p,
                                        ctypes.c_void_p]),
    "larva_l1_partial_grad": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_float, ctypes.c_float, _c_float_p,
                                             _c_int_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p]),
    "larva_l1_partial_grad_batch": (ctypes.c_int, [_c_pp, _c_float_p, ctypes.c_int, ctypes.c_float, ctypes.c_float, _c_pp,
                                                   _c_int_p, _c_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_void_p]),
    "larva_loss_from_partials_to_host": (ctypes.c_int, [_c_pp, _c_int_p, _c_float_p, ctypes.c_int, ctypes.c_float,
                                                        _c_float_p, _c_float_p, ctypes.c_void_p]),
    "larva_loss_from_partials_to_host_seq": (ctypes.c_int, [_c_pp, _c_int_p, _c_float_p, ctypes.c_int, ctypes.c_float,
                                                            _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "larva_host_cell_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "larva_host_cell_free": (ctypes.c_int, [_c_float_p]),
    "larva_loss_from_partials": (ctypes.c_int, [_c_pp, _c_int_p, _c_float_p, ctypes.c_int, ctypes.c_float,
                                                _c_float_p, ctypes.c_void_p]),
    "larva_sum_scalars": (ctypes.c_int, [_c_pp, ctypes.c_int, ctypes.c_float, _c_float_p, ctypes.c_void_p]),
    "larva_pixel_unshuffle4": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_wgrad_cu_share": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "larva_adamw_step_host": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_float, ctypes.c_longlong, ctypes.c_void_p]),
    "larva_gather_patches": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "larva_sqerr_u8": (ctypes.c_int, [_c_float_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "larva_adamw_step": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_float, ctypes.c_longlong, ctypes.c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle. Raises RuntimeError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # LARVA_HIP_LIB: another build of the same sources (same-box A/B timing of compile-time options)
    path = os.environ.get("LARVA_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "larvanet_amd: %s is missing. Build it with `python -m larvanet_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback." % path)
    # torch must own the process's HIP runtime: loading this library first would bring in a
    # second libamdhip64 that later finds "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        lib = load()
        msg = lib.larva_error_string(code)
        raise RuntimeError("larvanet_amd: %s failed: hip error %d (%s)" %
                           (what, code, msg.decode() if msg else "?"))


def ptr_array(ptrs):
    """Array of device pointers (ints / None) for a `const float* const*` argument."""
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)
