"""Tensor-level wrappers over the C ABI (hip_lib): torch supplies device memory and the stream,
every byte of arithmetic happens in liblarva_hip.so.  Each wrapper validates what the kernel and
its grid assume (device, dtype, contiguity, shapes) before launching.
"""
import ctypes
import os
import struct

import torch

from . import hip_lib

_SUPPORTED_COUT = (32, 48, 64)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("larvanet_amd: %s must be a tensor on a HIP device (no CPU path exists)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("larvanet_amd: %s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("larvanet_amd: %s must be contiguous" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError("larvanet_amd: %s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.data_ptr()


def _opt(t, name, shape):
    return None if t is None else _chk(t, name, shape)


def packed_weight_floats(cout, cin):
    return int(hip_lib.load().larva_packed_weight_floats(cout, cin))


def pack_weights(w, cin_off=0, cin=None, cin_pad=None, want_bwd=True):
    """w: [cout][cin_total][3][3] -> (wpk_fwd, wpk_bwd) packed images for the conv kernel.

    Packs the input-channel slice [cin_off, cin_off+cin) (default: to the end).  cin_pad pads
    with zero channels up to that kernel channel count (head conv: 3 -> 16); padding is only
    meaningful when the slice ends at cin_total."""
    lib = hip_lib.load()
    _chk(w, "w")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise RuntimeError("larvanet_amd: only [cout][cin][3][3] weights are supported")
    cout, cin_total = int(w.shape[0]), int(w.shape[1])
    cin = cin_total - cin_off if cin is None else cin
    if cin_off < 0 or cin < 1 or cin_off + cin > cin_total:
        raise RuntimeError("larvanet_amd: weight slice out of range")
    cin_k = cin if cin_pad is None else cin_pad
    if cin_k > cin and cin_off + cin != cin_total:
        raise RuntimeError("larvanet_amd: zero padding needs a slice that ends at cin_total")
    if cin_k % 8 or cout % 8 or cin_k < cin:
        raise RuntimeError("larvanet_amd: channel counts must be multiples of 8 (cout=%d cin=%d)" % (cout, cin_k))
    fwd = torch.empty(packed_weight_floats(cout, cin_k), device=w.device, dtype=torch.float32)
    bwd = torch.empty(packed_weight_floats(cin_k, cout), device=w.device, dtype=torch.float32) if want_bwd else None
    code = lib.larva_pack_weights(w.data_ptr(), fwd.data_ptr(), bwd.data_ptr() if want_bwd else None,
                                  cout, cin_k, cin_total, cin_off, _stream())
    hip_lib.check(code, "larva_pack_weights")
    return fwd, bwd


def pack_weights_batch(jobs):
    """jobs: list of (w, fwd_buf, bwd_buf or None, cout, cin_k, cin_off): packs every weight
    (slice) into its persistent kernel-layout buffers with one launch per 64 jobs."""
    lib = hip_lib.load()
    for i in range(0, len(jobs), 64):
        chunk = jobs[i:i + 64]
        ws, fs, bs, couts, cins, totals, offs = [], [], [], [], [], [], []
        for (w, fwd, bwd, cout, cin_k, cin_off) in chunk:
            _chk(w, "w")
            if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or int(w.shape[0]) != cout:
                raise RuntimeError("larvanet_amd: only [cout][cin][3][3] weights are supported")
            ws.append(w.data_ptr())
            fs.append(_chk(fwd, "wpk_fwd", (packed_weight_floats(cout, cin_k),)))
            bs.append(None if bwd is None else _chk(bwd, "wpk_bwd", (packed_weight_floats(cin_k, cout),)))
            couts.append(cout)
            cins.append(cin_k)
            totals.append(int(w.shape[1]))
            offs.append(cin_off)
        code = lib.larva_pack_weights_batch(hip_lib.ptr_array(ws), hip_lib.ptr_array(fs), hip_lib.ptr_array(bs),
                                            hip_lib.int_array(couts), hip_lib.int_array(cins),
                                            hip_lib.int_array(totals), hip_lib.int_array(offs), len(chunk), _stream())
        hip_lib.check(code, "larva_pack_weights_batch")


_STRIP_TABLES = {}


def strip_tile_table(H, P, device, phase=0):
    """Device copy of the library's strip-tile table of one H x P image (cached per shape and
    device; built and uploaded outside any stream capture) -> (tensor, tiles per image) or None when
    the height cannot be cut into 5- and 4-row tiles."""
    key = (int(H), int(P), str(device), int(phase))
    hit = _STRIP_TABLES.get(key)
    if hit is None:
        lib = hip_lib.load()
        cap = 4 * ((H + 3) // 4) * ((P + 15) // 16) + 16
        buf = (ctypes.c_uint * cap)()
        n = int(lib.larva_strip_tile_table(int(H), int(P), int(phase), buf, cap))
        if n <= 0 or n > cap:
            hit = False
        else:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("larvanet_amd: strip-tile table for %dx%d requested during stream capture "
                                   "(run the step once outside the capture first)" % (H, P))
            import numpy as np
            # (entries use bit 31: the raw 32-bit patterns travel as int32); the host array is kept: the launch passes
            # it along, and small tables then travel inside the kernel arguments (larva_conv3x3_fwd_strips)
            hit = (torch.from_numpy(np.frombuffer(buf, dtype=np.int32, count=n).copy()).to(device), n, buf)
        _STRIP_TABLES[key] = hit
    return hit or None


def step_prologue(jobs, x, x16, base):
    """pack_weights_batch(jobs) (<= 64 jobs) + x16[:, :C] = x + base = bicubic4(x) in one launch."""
    lib = hip_lib.load()
    if len(jobs) > 64:
        raise RuntimeError("larvanet_amd: at most 64 pack jobs per prologue launch")
    N, C, H, W = (int(v) for v in x.shape)
    _chk(x, "x")
    _chk(x16, "x16", (N, 16, H, W))
    _chk(base, "base", (N, C, 4 * H, 4 * W))
    ws, fs, bs, couts, cins, totals, offs = [], [], [], [], [], [], []
    for (w, fwd, bwd, cout, cin_k, cin_off) in jobs:
        _chk(w, "w")
        ws.append(w.data_ptr())
        fs.append(_chk(fwd, "wpk_fwd", (packed_weight_floats(cout, cin_k),)))
        bs.append(None if bwd is None else _chk(bwd, "wpk_bwd", (packed_weight_floats(cin_k, cout),)))
        couts.append(cout)
        cins.append(cin_k)
        totals.append(int(w.shape[1]))
        offs.append(cin_off)
    code = lib.larva_step_prologue(hip_lib.ptr_array(ws), hip_lib.ptr_array(fs), hip_lib.ptr_array(bs),
                                   hip_lib.int_array(couts), hip_lib.int_array(cins), hip_lib.int_array(totals),
                                   hip_lib.int_array(offs), len(jobs), x.data_ptr(), x16.data_ptr(), base.data_ptr(),
                                   N, C, H, W, _stream())
    hip_lib.check(code, "larva_step_prologue")


def conv3x3(srcs, wpk, cout, bias=None, relu=False, mask=None, res0=None, res1=None,
            shuffle=False, base=None, out=None, logical_w=None, images=None, strips=False, plain_stores=False,
            tile_rows=0):
    """Fused 3x3 conv over the channel concatenation of `srcs` (list of [N][c][H][P]).

    shuffle=False: returns [N][cout][H][P]; shuffle=True: returns PixelShuffle(4) layout
    [N][cout/16][4H][4W] (+ base).  logical_w: the image is W = logical_w <= P columns wide and
    the tensors' rows are padded to the pitch P (columns [W, P) hold zeros) -- lets widths that
    are not a multiple of 4 use the 16-byte staging path.
    images=(lo, hi): only images [lo, hi) of the batch are computed (every operand is the full-batch
    tensor; the other images of `out` are left untouched).  strips=True (or 2: the tile table
    starts with the other tile height): 5 x 16 / 4 x 16 tiles instead of 3 x 48 (same results bit for
    bit; see larva_conv3x3_fwd_strips) where the shape allows, else the regular tiles; plain_stores:
    the strip launch writes its output with plain instead of non-temporal stores.
    tile_rows: 0 = the library picks the tiling of a whole-tensor launch (3 x 48 tiles; 4 x 48 for large 32-channel
    launches; persistent workgroups where there are more tiles than workgroup slots), 3 / 4 = that tile height (tests,
    A/B timing; larva_conv3x3_fwd_tiled)."""
    lib = hip_lib.load()
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    if not 1 <= len(srcs) <= 8:
        raise RuntimeError("larvanet_amd: 1..8 source tensors")
    if cout not in _SUPPORTED_COUT:
        raise RuntimeError("larvanet_amd: cout must be one of %s" % (_SUPPORTED_COUT,))
    N, cps, H, P = (int(v) for v in srcs[0].shape)
    W = P if logical_w is None else int(logical_w)
    if not 0 < W <= P:
        raise RuntimeError("larvanet_amd: logical width %d does not fit the row pitch %d" % (W, P))
    if cps % 8:
        raise RuntimeError("larvanet_amd: input channels per tensor must be a multiple of 8")
    ptrs = [_chk(s, "src[%d]" % i, (N, cps, H, P)) for i, s in enumerate(srcs)]
    cin = cps * len(srcs)
    _chk(wpk, "wpk", (packed_weight_floats(cout, cin),))
    full = (N, cout, H, P)
    hr = (N, cout // 16, 4 * H, 4 * W)
    if out is None:
        out = torch.empty(hr if shuffle else full, device=srcs[0].device, dtype=torch.float32)
    _chk(out, "out", hr if shuffle else full)
    lo, hi = (0, N) if images is None else (int(images[0]), int(images[1]))
    if not 0 <= lo < hi <= N:
        raise RuntimeError("larvanet_amd: image range [%d, %d) outside the batch of %d" % (lo, hi, N))

    def at(ptr, per_image_floats):   # the same operand, starting at image `lo`
        return None if ptr is None else ptr + 4 * lo * per_image_floats

    lr_img, hr_img = H * P, 16 * H * W
    args = (hip_lib.ptr_array([at(p, cps * lr_img) for p in ptrs]), len(srcs), cps, wpk.data_ptr(),
            _opt(bias, "bias", (cout,)), at(_opt(res0, "res0", full), cout * lr_img),
            at(_opt(res1, "res1", full), cout * lr_img), at(_opt(mask, "mask", full), cout * lr_img),
            at(_opt(base, "base", hr), (cout // 16) * hr_img),
            at(out.data_ptr(), (cout // 16) * hr_img if shuffle else cout * lr_img),
            hi - lo, cout, H, W, P, 1 if relu else 0, 1 if shuffle else 0)
    if strips and cout in (48, 32, 64):
        tab = strip_tile_table(H, P, out.device, phase=1 if strips == 2 else 0)
        if tab is not None:
            code = lib.larva_conv3x3_fwd_strips(*args, tab[0].data_ptr(), tab[2], tab[1], 1 if plain_stores else 0, _stream())
            if code != 801:   # hipErrorNotSupported: unaligned operands -> the regular tiles below
                hip_lib.check(code, "larva_conv3x3_fwd_strips")
                return out
    if tile_rows:
        code = lib.larva_conv3x3_fwd_tiled(*args, int(tile_rows), _stream())
    else:
        code = lib.larva_conv3x3_fwd_pitched(*args, _stream())
    hip_lib.check(code, "larva_conv3x3_fwd")
    return out


def conv3x3_batch(jobs, cout, relu=False, shuffle=False, logical_w=None):
    """2..4 INDEPENDENT convs of one shape and one fusion in one launch (their workgroups share
    the CUs two by two).  jobs: dicts with srcs (tensor or list), wpk and optionally bias, mask,
    res0, res1, base -- a fusion operand is given by every job or by none.  Returns the outputs.
    Falls back to one launch per job where the 16-byte staging path does not apply."""
    lib = hip_lib.load()
    if not 2 <= len(jobs) <= 4:
        raise RuntimeError("larvanet_amd: 2..4 conv jobs per batched launch")
    norm = []
    for j in jobs:
        srcs = j["srcs"]
        norm.append(dict(j, srcs=[srcs] if isinstance(srcs, torch.Tensor) else list(srcs)))
    n_src = len(norm[0]["srcs"])
    N, cps, H, P = (int(v) for v in norm[0]["srcs"][0].shape)
    W = P if logical_w is None else int(logical_w)
    if cout not in _SUPPORTED_COUT or cps % 8 or not 0 < W <= P or not 1 <= n_src <= 8:
        raise RuntimeError("larvanet_amd: unsupported batched conv shape")
    cin = cps * n_src
    full, hr = (N, cout, H, P), (N, cout // 16, 4 * H, 4 * W)
    names = ("bias", "res0", "res1", "mask", "base")
    used = {k: norm[0].get(k) is not None for k in names}
    src_ptrs, cols, outs = [], {k: [] for k in names}, []
    for j in norm:
        if len(j["srcs"]) != n_src or any((j.get(k) is not None) != used[k] for k in names):
            raise RuntimeError("larvanet_amd: batched conv jobs must share one shape and one fusion")
        src_ptrs += [_chk(t, "src", (N, cps, H, P)) for t in j["srcs"]]
        _chk(j["wpk"], "wpk", (packed_weight_floats(cout, cin),))
        for k, shape in (("bias", (cout,)), ("res0", full), ("res1", full), ("mask", full), ("base", hr)):
            if used[k]:
                cols[k].append(_chk(j[k], k, shape))
        outs.append(torch.empty(hr if shuffle else full, device=j["srcs"][0].device, dtype=torch.float32))

    def arr(k):
        return hip_lib.ptr_array(cols[k]) if used[k] else None

    common = (len(norm), hip_lib.ptr_array(src_ptrs), n_src, cps, hip_lib.ptr_array([j["wpk"].data_ptr() for j in norm]),
              arr("bias"), arr("res0"), arr("res1"), arr("mask"), arr("base"),
              hip_lib.ptr_array([o.data_ptr() for o in outs]), N, cout, H, W, P, 1 if relu else 0, 1 if shuffle else 0)
    code = lib.larva_conv3x3_fwd_batch(*common, _stream())
    if code == 801:  # hipErrorNotSupported: unaligned shape, one launch per job
        return [conv3x3(j["srcs"], j["wpk"], cout, bias=j.get("bias"), relu=relu, mask=j.get("mask"),
                        res0=j.get("res0"), res1=j.get("res1"), shuffle=shuffle, base=j.get("base"), out=o,
                        logical_w=logical_w)
                for j, o in zip(norm, outs)]
    hip_lib.check(code, "larva_conv3x3_fwd_batch")
    return outs


def head_conv3_direct(x, w, bias=None, pitch=None):
    """nn.Conv2d(3, cout, 3, 1, 1) on the raw image: x [N][3][H][W], w [cout][3][3][3] -> [N][cout][H][P]
    (P = pitch or W; columns [W, P) zero)."""
    lib = hip_lib.load()
    N, C, H, W = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    if C != 3 or tuple(w.shape[1:]) != (3, 3, 3) or cout % 16:
        raise RuntimeError("larvanet_amd: the direct head conv takes a 3-channel image and [cout][3][3][3] weights")
    _chk(x, "x")
    _chk(w, "w")
    P = W if pitch is None else int(pitch)
    out = torch.empty((N, cout, H, P), device=x.device, dtype=torch.float32)
    hip_lib.check(lib.larva_head_conv3_direct(x.data_ptr(), w.data_ptr(), _opt(bias, "bias", (cout,)), out.data_ptr(),
                                              N, cout, H, W, P, _stream()), "larva_head_conv3_direct")
    return out


def conv3x3_exit_l1_batch(jobs, cout, truth, gvalue, gscale, want_image):
    """2..4 exits scored by L1 inside their pixel-shuffle conv launch.  jobs: dicts {srcs, wpk, bias,
    base}; truth [N][cout/16][4H][4W]; the gradient of every image element is sign(out - truth) *
    (gvalue * gscale) / numel; want_image[j]: also store exit j's image.  Returns (images (None where
    not wanted), partial-sum vectors, gradients [N][cout][H][W]) or None when the launch does not
    apply (unaligned operands, cout != 48): the caller then runs the conv and the L1 sweep separately."""
    lib = hip_lib.load()
    if not 2 <= len(jobs) <= 4 or cout != 48:
        return None
    norm = [dict(j, srcs=[j["srcs"]] if isinstance(j["srcs"], torch.Tensor) else list(j["srcs"])) for j in jobs]
    n_src = len(norm[0]["srcs"])
    N, cps, H, P = (int(v) for v in norm[0]["srcs"][0].shape)
    hr = (N, cout // 16, 4 * H, 4 * P)
    _chk(truth, "truth", hr)
    numel = float(truth.numel())
    import numpy as np
    gval = float((np.float32(gvalue) * np.float32(gscale)) * (np.float32(1.0) / np.float32(numel)))
    npart = int(lib.larva_exit_l1_partials(N, H, P))
    src_ptrs, outs, grads, parts = [], [], [], []
    for j, want in zip(norm, want_image):
        if len(j["srcs"]) != n_src:
            raise RuntimeError("larvanet_amd: batched exits must share one shape")
        src_ptrs += [_chk(t, "src", (N, cps, H, P)) for t in j["srcs"]]
        _chk(j["wpk"], "wpk", (packed_weight_floats(cout, cps * n_src),))
        _chk(j["bias"], "bias", (cout,))
        _chk(j["base"], "base", hr)
        outs.append(torch.empty(hr, device=truth.device, dtype=torch.float32) if want else None)
        grads.append(torch.empty((N, cout, H, P), device=truth.device, dtype=torch.float32))
        parts.append(torch.empty(npart, device=truth.device, dtype=torch.float32))
    code = lib.larva_conv3x3_exit_l1_batch(
        len(norm), hip_lib.ptr_array(src_ptrs), n_src, cps, hip_lib.ptr_array([j["wpk"].data_ptr() for j in norm]),
        hip_lib.ptr_array([j["bias"].data_ptr() for j in norm]), hip_lib.ptr_array([j["base"].data_ptr() for j in norm]),
        hip_lib.ptr_array([truth.data_ptr()] * len(norm)), hip_lib.ptr_array([None if o is None else o.data_ptr() for o in outs]),
        hip_lib.ptr_array([g.data_ptr() for g in grads]), hip_lib.ptr_array([p.data_ptr() for p in parts]),
        gval, N, cout, H, P, P, _stream())
    if code == 801:
        return None
    hip_lib.check(code, "larva_conv3x3_exit_l1_batch")
    return outs, parts, grads


def wgrad_partial_floats(cout, cin, splits):
    return int(hip_lib.load().larva_wgrad_partial_floats(cout, cin, splits))


def max_wgrad_jobs():
    return 64


def wgrad_cu_share(cout, cin):
    """Workgroups of the (cout, cin) weight-gradient kernel that fit one CU together (1 or 2)."""
    return int(hip_lib.load().larva_wgrad_cu_share(int(cout), int(cin)))


def conv3x3_wgrad(jobs, cout, cin, splits):
    """jobs: list (<= max_wgrad_jobs()) of dicts {dy, x, dw, db (or None), cin_off, cin_valid}; dw/db are
    overwritten.  All jobs share (N, cout, cin, H, W)."""
    lib = hip_lib.load()
    if not 1 <= len(jobs) <= max_wgrad_jobs():
        raise RuntimeError("larvanet_amd: 1..%d wgrad jobs per call" % max_wgrad_jobs())
    N, _, H, W = (int(v) for v in jobs[0]["dy"].shape)
    dys, xs, parts, dws, dbs, offs, valids, totals, keep = [], [], [], [], [], [], [], [], []
    nfl = wgrad_partial_floats(cout, cin, splits)
    for j in jobs:
        dys.append(_chk(j["dy"], "dy", (N, cout, H, W)))
        xs.append(_chk(j["x"], "x", (N, cin, H, W)))
        dw = j["dw"]
        _chk(dw, "dw")
        if dw.dim() != 4 or int(dw.shape[0]) != cout or tuple(dw.shape[2:]) != (3, 3):
            raise RuntimeError("larvanet_amd: dw must be [cout][cin_total][3][3]")
        total = int(dw.shape[1])
        off = int(j.get("cin_off", 0))
        valid = int(j.get("cin_valid", cin))
        if off < 0 or valid < 1 or valid > cin or off + valid > total:
            raise RuntimeError("larvanet_amd: wgrad channel slice out of range")
        part = j.get("partial")
        if part is None:
            part = torch.empty(nfl, device=dw.device, dtype=torch.float32)
        _chk(part, "partial", (nfl,))
        keep.append(part)
        parts.append(part.data_ptr())
        dws.append(dw.data_ptr())
        dbs.append(_opt(j.get("db"), "db", (cout,)))
        offs.append(off)
        valids.append(valid)
        totals.append(total)
    code = lib.larva_conv3x3_wgrad(
        hip_lib.ptr_array(dys), hip_lib.ptr_array(xs), hip_lib.ptr_array(parts), hip_lib.ptr_array(dws),
        hip_lib.ptr_array(dbs), hip_lib.int_array(offs), hip_lib.int_array(valids), hip_lib.int_array(totals),
        len(jobs), splits, N, cout, cin, H, W, _stream())
    hip_lib.check(code, "larva_conv3x3_wgrad")
    return keep


def conv3x3_wgrad_partial(jobs, cout, cin, splits):
    """Phase 1 only: jobs (<= 64 dicts {dy, x}) -> (partial tensors, splits actually used).
    Feed them to wgrad_reduce later (the partials must stay alive until then)."""
    lib = hip_lib.load()
    if not 1 <= len(jobs) <= 64:
        raise RuntimeError("larvanet_amd: 1..64 wgrad jobs per call")
    N, _, H, W = (int(v) for v in jobs[0]["dy"].shape)
    splits = max(1, min(int(splits), N * ((H + 2) // 3) * ((W + 47) // 48)))  # the library's clamp
    nfl = wgrad_partial_floats(cout, cin, splits)
    dys = [_chk(j["dy"], "dy", (N, cout, H, W)) for j in jobs]
    xs = [_chk(j["x"], "x", (N, cin, H, W)) for j in jobs]
    parts = [torch.empty(nfl, device=jobs[0]["dy"].device, dtype=torch.float32) for _ in jobs]
    used = ctypes.c_int(0)
    code = lib.larva_conv3x3_wgrad_partial(
        hip_lib.ptr_array(dys), hip_lib.ptr_array(xs), hip_lib.ptr_array([p.data_ptr() for p in parts]),
        len(jobs), splits, N, cout, cin, H, W, ctypes.byref(used), _stream())
    hip_lib.check(code, "larva_conv3x3_wgrad_partial")
    return parts, int(used.value)


def conv3x3_wgrad_partial_flat(jobs, cout, cin, nwg, head=None):
    """Phase 1 of ALL jobs (<= 64 dicts {dy, x}; 48 -> 48, 32 -> 32 or 64 -> 64 channels) as one grid of `nwg` workgroups ->
    (partial tensors, [partial images per job]) or None when the flat launch does not apply.  head: one more dict
    {dy [N][48][H][W], x [N][16][H][W]} -- the 3 -> 48 head on its padded input -- whose tiles the last workgroups of
    the same grid take; the result then ends with that job's partial tensor / image count."""
    lib = hip_lib.load()
    if not 1 <= len(jobs) <= 64 or (cout, cin) not in ((48, 48), (32, 32), (64, 64)) or (head is not None and cout != 48):
        return None
    N, _, H, W = (int(v) for v in jobs[0]["dy"].shape)
    if W % 4:
        return None
    tiles = N * ((H + 2) // 3) * ((W + 47) // 48)
    cap = int(lib.larva_wgrad_flat_max_splits(len(jobs), int(nwg), tiles))
    nfl = wgrad_partial_floats(cout, cin, cap)
    dys = [_chk(j["dy"], "dy", (N, cout, H, W)) for j in jobs]
    xs = [_chk(j["x"], "x", (N, cin, H, W)) for j in jobs]
    dev = jobs[0]["dy"].device
    parts = [torch.empty(nfl, device=dev, dtype=torch.float32) for _ in jobs]
    used = (ctypes.c_int * len(jobs))()
    if head is None:
        code = lib.larva_conv3x3_wgrad_partial_flat(
            hip_lib.ptr_array(dys), hip_lib.ptr_array(xs), hip_lib.ptr_array([p.data_ptr() for p in parts]),
            len(jobs), int(nwg), N, cout, cin, H, W, used, _stream())
    else:
        hcap = int(lib.larva_wgrad_flat_head_splits(len(jobs), int(nwg), tiles))
        hper = wgrad_partial_floats(48, 16, 1)
        hpart = torch.empty(hper * hcap, device=dev, dtype=torch.float32)
        hused = ctypes.c_int(0)
        code = lib.larva_conv3x3_wgrad_partial_flat_head(
            hip_lib.ptr_array(dys), hip_lib.ptr_array(xs), hip_lib.ptr_array([p.data_ptr() for p in parts]), len(jobs),
            _chk(head["dy"], "head dy", (N, 48, H, W)), _chk(head["x"], "head x", (N, 16, H, W)), hpart.data_ptr(),
            int(nwg), N, H, W, used, ctypes.byref(hused), _stream())
    if code == 801:
        return None
    hip_lib.check(code, "larva_conv3x3_wgrad_partial_flat")
    splits = [int(v) for v in used]
    per = wgrad_partial_floats(cout, cin, 1)
    out = [p[:per * s] for p, s in zip(parts, splits)]
    if head is not None:
        out.append(hpart[:hper * int(hused.value)])
        splits.append(int(hused.value))
    return out, splits


def _loss_terms(terms, scales):
    ptrs, counts = [], []
    for t in terms:
        _chk(t, "term")
        if t.dim() > 1:
            raise RuntimeError("larvanet_amd: a loss term is a scalar or a vector of partial sums")
        ptrs.append(t.data_ptr())
        counts.append(max(1, int(t.numel())))
    sc = (ctypes.c_float * len(terms))(*[float(v) for v in scales])
    return hip_lib.ptr_array(ptrs), hip_lib.int_array(counts), sc


def wgrad_reduce(jobs, cout=None, cin=None, loss=None):
    """Phase 2: jobs (<= 64 dicts {partial, splits, dw, db (or None), cin_off, cin_valid[, cout, cin]})
    reduced in ONE launch; dw/db are overwritten.  cout/cin: the kernel shape of every job that does
    not carry its own.  loss = (terms, scales, divisor, out): the same launch also finishes
    loss_from_partials(terms, scales, divisor) into the 0-d tensor `out`."""
    lib = hip_lib.load()
    if not 1 <= len(jobs) <= 64:
        raise RuntimeError("larvanet_amd: 1..64 reduce jobs per call")
    parts, dws, dbs, offs, valids, totals, splits, couts, cins = [], [], [], [], [], [], [], [], []
    for j in jobs:
        co, ci = int(j.get("cout", cout)), int(j.get("cin", cin))
        dw = j["dw"]
        _chk(dw, "dw")
        if dw.dim() != 4 or int(dw.shape[0]) != co or tuple(dw.shape[2:]) != (3, 3):
            raise RuntimeError("larvanet_amd: dw must be [cout][cin_total][3][3]")
        total = int(dw.shape[1])
        off = int(j.get("cin_off", 0))
        valid = int(j.get("cin_valid", ci))
        if off < 0 or valid < 1 or valid > ci or off + valid > total:
            raise RuntimeError("larvanet_amd: wgrad channel slice out of range")
        sp = int(j["splits"])
        parts.append(_chk(j["partial"], "partial", (wgrad_partial_floats(co, ci, sp),)))
        dws.append(dw.data_ptr())
        dbs.append(_opt(j.get("db"), "db", (co,)))
        offs.append(off)
        valids.append(valid)
        totals.append(total)
        splits.append(sp)
        couts.append(co)
        cins.append(ci)
    common = (hip_lib.ptr_array(parts), hip_lib.ptr_array(dws), hip_lib.ptr_array(dbs), hip_lib.int_array(offs),
              hip_lib.int_array(valids), hip_lib.int_array(totals), hip_lib.int_array(splits), hip_lib.int_array(couts),
              hip_lib.int_array(cins), len(jobs))
    if loss is None:
        code = lib.larva_wgrad_reduce(*common, _stream())
    else:
        terms, scales, divisor, out = loss
        if not 1 <= len(terms) <= 8:
            raise RuntimeError("larvanet_amd: 1..8 loss terms")
        _chk(out, "loss", ())
        ptrs, counts, sc = _loss_terms(terms, scales)
        code = lib.larva_wgrad_reduce_with_loss(*common, ptrs, counts, sc, len(terms), float(divisor), out.data_ptr(), _stream())
    hip_lib.check(code, "larva_wgrad_reduce")


UPSAMPLE_MODES = {"bicubic": 0, "bilinear": 1}


def upsample4(x, mode="bicubic"):
    """F.interpolate(x, scale_factor=4, mode=mode, align_corners=False) (models/LarvaNet.py:283-285) for the two
    modes that call accepts for a 4-D input (nearest / area refuse align_corners, linear / trilinear the rank)."""
    lib = hip_lib.load()
    if mode not in UPSAMPLE_MODES:
        raise RuntimeError("larvanet_amd: no x4 kernel for interpolate mode %r" % (mode,))
    N, C, H, W = (int(v) for v in x.shape)
    _chk(x, "x")
    out = torch.empty((N, C, 4 * H, 4 * W), device=x.device, dtype=torch.float32)
    hip_lib.check(lib.larva_upsample4_fwd(x.data_ptr(), out.data_ptr(), N, C, H, W, UPSAMPLE_MODES[mode], _stream()),
                  "larva_upsample4_fwd")
    return out


def bicubic4(x):
    lib = hip_lib.load()
    N, C, H, W = (int(v) for v in x.shape)
    _chk(x, "x")
    out = torch.empty((N, C, 4 * H, 4 * W), device=x.device, dtype=torch.float32)
    hip_lib.check(lib.larva_bicubic4_fwd(x.data_ptr(), out.data_ptr(), N, C, H, W, _stream()), "larva_bicubic4_fwd")
    return out


_L1_WORKSPACES = {}


def _l1_workspace(device):
    """Block partial sums of l1_fwd, one buffer per (device, stream)."""
    lib = hip_lib.load()
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    ws = _L1_WORKSPACES.get(key)
    if ws is None:
        ws = torch.empty(int(lib.larva_l1_workspace_floats()), device=device, dtype=torch.float32)
        _L1_WORKSPACES[key] = ws
    return ws


def l1_fwd(a, b):
    """mean |a - b| as a 0-d device tensor (one launch)."""
    lib = hip_lib.load()
    _chk(a, "a")
    _chk(b, "b", a.shape)
    ws = _l1_workspace(a.device)
    loss = torch.empty((), device=a.device, dtype=torch.float32)
    hip_lib.check(lib.larva_l1_fwd(a.data_ptr(), b.data_ptr(), a.numel(), ws.data_ptr(), loss.data_ptr(), _stream()),
                  "larva_l1_fwd")
    return loss


def l1_partial(a, b):
    """Block partial sums of sum|a - b| -> (partials [blocks], 1 / numel): an L1 term that
    loss_from_partials finishes together with the other exits' terms."""
    lib = hip_lib.load()
    _chk(a, "a")
    _chk(b, "b", a.shape)
    part = torch.empty(int(lib.larva_l1_workspace_floats()), device=a.device, dtype=torch.float32)
    blocks = ctypes.c_int(0)
    hip_lib.check(lib.larva_l1_partial(a.data_ptr(), b.data_ptr(), a.numel(), part.data_ptr(), ctypes.byref(blocks),
                                       _stream()), "larva_l1_partial")
    return part[:int(blocks.value)], 1.0 / float(a.numel())


def l1_partial_grad(a, b, gvalue, gscale):
    """l1_partial + l1_bwd_unshuffle4 in one pass, for an upstream gradient known on the host:
    -> (partials, 1 / numel, grad [N][16C][H][W])."""
    lib = hip_lib.load()
    _chk(a, "a")
    _chk(b, "b", a.shape)
    N, C, HH, WW = (int(v) for v in a.shape)
    if HH % 4 or WW % 4:
        raise RuntimeError("larvanet_amd: spatial dims must be divisible by 4")
    part = torch.empty(int(lib.larva_l1_workspace_floats()), device=a.device, dtype=torch.float32)
    grad = torch.empty((N, 16 * C, HH // 4, WW // 4), device=a.device, dtype=torch.float32)
    blocks = ctypes.c_int(0)
    hip_lib.check(lib.larva_l1_partial_grad(a.data_ptr(), b.data_ptr(), float(gvalue), float(gscale), part.data_ptr(),
                                            ctypes.byref(blocks), grad.data_ptr(), N, C, HH // 4, WW // 4, _stream()),
                  "larva_l1_partial_grad")
    return part[:int(blocks.value)], 1.0 / float(a.numel()), grad


def l1_partial_grad_batch(outs, truth, gvalue, gscale):
    """l1_partial_grad for several exit images against one truth in one launch ->
    ([partials], 1 / numel, [grads])."""
    lib = hip_lib.load()
    if not 1 <= len(outs) <= 8:
        raise RuntimeError("larvanet_amd: 1..8 images per batched L1 sweep")
    _chk(truth, "truth")
    N, C, HH, WW = (int(v) for v in truth.shape)
    if HH % 4 or WW % 4:
        raise RuntimeError("larvanet_amd: spatial dims must be divisible by 4")
    ptrs = [_chk(o, "out", truth.shape) for o in outs]
    nws = int(lib.larva_l1_workspace_floats())
    parts = [torch.empty(nws, device=truth.device, dtype=torch.float32) for _ in outs]
    grads = [torch.empty((N, 16 * C, HH // 4, WW // 4), device=truth.device, dtype=torch.float32) for _ in outs]
    blocks = ctypes.c_int(0)
    hip_lib.check(lib.larva_l1_partial_grad_batch(
        hip_lib.ptr_array(ptrs), truth.data_ptr(), len(outs), float(gvalue), float(gscale),
        hip_lib.ptr_array([p.data_ptr() for p in parts]), ctypes.byref(blocks),
        hip_lib.ptr_array([g.data_ptr() for g in grads]), N, C, HH // 4, WW // 4, _stream()), "larva_l1_partial_grad_batch")
    nb = int(blocks.value)
    return [p[:nb] for p in parts], 1.0 / float(truth.numel()), grads


class HostCell:
    """{float value, uint32 sequence} in coherent pinned host memory a kernel can store into (larva_host_cell_alloc):
    the host reads it without synchronising with a stream.  Every store of a kernel bumps the sequence number; the
    owner counts its launches (expect()) and take() returns a value only once the launch it waits for has stored."""

    def __init__(self):
        p = ctypes.c_void_p()
        hip_lib.check(hip_lib.load().larva_host_cell_alloc(ctypes.byref(p)), "larva_host_cell_alloc")
        self.ptr = int(p.value)
        self._cell = ctypes.c_uint64.from_address(self.ptr)
        self.expected = 0
        # the sequence number in device memory as well: the storing launch reads it there instead of over PCIe
        self.dev_seq = torch.zeros(1, dtype=torch.int32, device="cuda") if torch.cuda.is_available() else None

    def expect(self):
        """Call once per launch (or graph replay) that stores into the cell, before it is issued."""
        self.expected = (self.expected + 1) & 0xFFFFFFFF

    def take(self):
        """The value of the launch announced last, or None while it has not stored yet."""
        raw = self._cell.value            # one aligned 8-byte load: value and sequence number belong together
        if (raw >> 32) != self.expected:
            return None
        return struct.unpack("<f", struct.pack("<I", raw & 0xFFFFFFFF))[0]

    @property
    def value(self):
        return struct.unpack("<f", struct.pack("<I", self._cell.value & 0xFFFFFFFF))[0]

    @property
    def sequence(self):
        return self._cell.value >> 32

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if ptr:
            try:
                hip_lib.load().larva_host_cell_free(ptr)   # (synchronises with the device first)
            except Exception:   # interpreter shutdown
                pass


def loss_from_partials(terms, scales, divisor, host_cell=None):
    """( sum_i scales[i] * terms[i].sum() ) / divisor as a 0-d tensor, one launch, fixed order; host_cell
    (HostCell) receives the value too."""
    lib = hip_lib.load()
    if not 1 <= len(terms) <= 8:
        raise RuntimeError("larvanet_amd: 1..8 loss terms")
    ptrs, counts = [], []
    for t in terms:
        _chk(t, "term")
        if t.dim() > 1:
            raise RuntimeError("larvanet_amd: a loss term is a scalar or a vector of partial sums")
        ptrs.append(t.data_ptr())
        counts.append(max(1, int(t.numel())))
    out = torch.empty((), device=terms[0].device, dtype=torch.float32)
    sc = (ctypes.c_float * len(terms))(*[float(v) for v in scales])
    hip_lib.check(lib.larva_loss_from_partials_to_host_seq(
        hip_lib.ptr_array(ptrs), hip_lib.int_array(counts), sc, len(terms), float(divisor), out.data_ptr(),
        host_cell.ptr if host_cell is not None else None,
        host_cell.dev_seq.data_ptr() if host_cell is not None and host_cell.dev_seq is not None else None, _stream()),
        "larva_loss_from_partials_to_host")
    return out


def l1_bwd_unshuffle4(a, b, gout, gscale=1.0):
    """Gradient of mean|a - b| * gscale w.r.t. a, written as [N][16C][H][W] (pixel-unshuffled)."""
    lib = hip_lib.load()
    _chk(a, "a")
    _chk(b, "b", a.shape)
    _chk(gout, "gout", ())
    N, C, HH, WW = (int(v) for v in a.shape)
    if HH % 4 or WW % 4:
        raise RuntimeError("larvanet_amd: spatial dims must be divisible by 4")
    out = torch.empty((N, 16 * C, HH // 4, WW // 4), device=a.device, dtype=torch.float32)
    hip_lib.check(lib.larva_l1_bwd_unshuffle4(a.data_ptr(), b.data_ptr(), gout.data_ptr(), float(gscale),
                                              out.data_ptr(), N, C, HH // 4, WW // 4, _stream()),
                  "larva_l1_bwd_unshuffle4")
    return out


def sum_scalars(terms, divisor):
    lib = hip_lib.load()
    ptrs = [_chk(t, "term", ()) for t in terms]
    out = torch.empty((), device=terms[0].device, dtype=torch.float32)
    hip_lib.check(lib.larva_sum_scalars(hip_lib.ptr_array(ptrs), len(ptrs), float(divisor), out.data_ptr(), _stream()),
                  "larva_sum_scalars")
    return out


def l1_bwd(a, b, gout):
    lib = hip_lib.load()
    _chk(a, "a")
    _chk(b, "b", a.shape)
    _chk(gout, "gout", ())
    ga = torch.empty_like(a)
    hip_lib.check(lib.larva_l1_bwd(a.data_ptr(), b.data_ptr(), gout.data_ptr(), a.numel(), ga.data_ptr(), _stream()),
                  "larva_l1_bwd")
    return ga


def pixel_unshuffle4(g):
    lib = hip_lib.load()
    _chk(g, "g")
    N, C, HH, WW = (int(v) for v in g.shape)
    if HH % 4 or WW % 4:
        raise RuntimeError("larvanet_amd: pixel_unshuffle4 needs spatial dims divisible by 4")
    H, W = HH // 4, WW // 4
    out = torch.empty((N, 16 * C, H, W), device=g.device, dtype=torch.float32)
    hip_lib.check(lib.larva_pixel_unshuffle4(g.data_ptr(), out.data_ptr(), N, C, H, W, _stream()),
                  "larva_pixel_unshuffle4")
    return out


def adamw_step(p, g, m, v, step_lr, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    lib = hip_lib.load()
    n = p.numel()
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name, (n,))
    _chk(step_lr, "step_lr", (2,))
    hip_lib.check(lib.larva_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), step_lr.data_ptr(),
                                       beta1, beta2, eps, weight_decay, grad_scale, n, _stream()),
                  "larva_adamw_step")


def gather_patches(data, offsets, hw, draws, batch, patch, mult, out=None):
    """Augmented float patches [batch][3][patch][patch] from a uint8 dataset resident on the device
    (into `out` when given: e.g. the input buffer a captured training step reads)."""
    lib = hip_lib.load()
    for t, name, dt in ((data, "data", torch.uint8), (offsets, "offsets", torch.int64), (hw, "hw", torch.int32),
                        (draws, "draws", torch.int32)):
        if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise RuntimeError("larvanet_amd: %s must be a contiguous %s tensor on the HIP device" % (name, dt))
    if tuple(draws.shape) != (batch, 5):
        raise RuntimeError("larvanet_amd: draws must be [batch][5]")
    if out is None:
        out = torch.empty((batch, 3, patch, patch), device=data.device, dtype=torch.float32)
    _chk(out, "out", (batch, 3, patch, patch))
    hip_lib.check(lib.larva_gather_patches(data.data_ptr(), offsets.data_ptr(), hw.data_ptr(), draws.data_ptr(),
                                           out.data_ptr(), batch, patch, mult, _stream()), "larva_gather_patches")
    return out


def adamw_step_host(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, copy=None):
    """One AdamW step over flat buffers, step count and learning rate given by the host.  copy = (src, dst):
    the launch also copies the 0-d tensor src into dst (the step's loss out of a captured graph's buffer)."""
    lib = hip_lib.load()
    n = p.numel()
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, name, (n,))
    if copy is not None:
        src, dst = copy
        _chk(src, "copy source", ())
        _chk(dst, "copy destination", ())
        hip_lib.check(lib.larva_adamw_step_host_copy(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), int(step),
                                                     float(lr), beta1, beta2, eps, weight_decay, grad_scale, n,
                                                     src.data_ptr(), dst.data_ptr(), _stream()), "larva_adamw_step_host_copy")
        return
    hip_lib.check(lib.larva_adamw_step_host(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), int(step),
                                            float(lr), beta1, beta2, eps, weight_decay, grad_scale, n, _stream()),
                  "larva_adamw_step_host")


def psnr_u8(out_chw, truth_u8):
    """RGB PSNR of validate.py:17-27 computed on the device: out_chw float [C][H][W] (device),
    truth_u8 uint8 [C][TH][TW] (device), truth cropped top-left to the output size.  Returns a float
    (one 8-byte read-back instead of the whole HR image)."""
    import math
    lib = hip_lib.load()
    _chk(out_chw, "out")
    if not truth_u8.is_cuda or truth_u8.dtype != torch.uint8 or not truth_u8.is_contiguous():
        raise RuntimeError("larvanet_amd: truth must be a contiguous uint8 tensor on the HIP device")
    C, H, W = (int(v) for v in out_chw.shape)
    TC, TH, TW = (int(v) for v in truth_u8.shape)
    if TC != C or TH < H or TW < W:
        raise RuntimeError("larvanet_amd: truth image smaller than the output")
    acc = torch.zeros(1, device=out_chw.device, dtype=torch.int64)
    hip_lib.check(lib.larva_sqerr_u8(out_chw.data_ptr(), truth_u8.data_ptr(), C, H, W, TH, TW, acc.data_ptr(),
                                     _stream()), "larva_sqerr_u8")
    sq = int(acc.item())
    mse = sq / float(C * H * W)
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 ** 2 / mse)
