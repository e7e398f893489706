"""Tests of the MEASUREMENT library (tools/_diag/diag.so: tools/build_diag.sh, tools/larva_diag.h, tools/csrc/): the timed
launches bench.py --extras uses and the two closed round-5 experiments (layer pipeline, one-launch pair chain).  They are
NOT product tests: marked `diag` only, so `pytest -m gpu` counts product tests alone.  Run them on a GPU
box with `pytest tests -m diag`; without a HIP device or without the library they skip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.diag


def _rand(rng, shape, scale):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def hip_device():
    if not torch.cuda.is_available():
        pytest.skip("measurement-library tests need a HIP device")
    from larvanet_amd import hip_lib
    hip_lib.load()
    return torch.device("cuda", 0)


def _diag_lib_or_skip():
    """tools/diag_lib.py: the binding of the measurement library (tools/build_diag.sh; __graft_entry__.build() builds it)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("diag_lib", os.path.join(root, "tools", "diag_lib.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    if not D.available():
        pytest.skip("tools/_diag/diag.so has not been built (tools/build_diag.sh diag)")
    return D


@pytest.mark.parametrize("N,H,W,bpm,modules,merge", [(1, 339, 510, 4, 4, False), (1, 339, 510, 2, 4, True), (2, 101, 250, 2, 2, False),
                                                     (1, 48, 48, 2, 1, False)])
def test_layer_pipeline_equals_the_per_layer_launches_bit_for_bit(hip_device, N, H, W, bpm, modules, merge):
    """Round 5: the body of a full-image forward (models/LarvaNet.py:205-220,236-248,283-293 as validate.py:94-102 calls it)
    as ONE launch of persistent workgroups that walk (layer, tile) positions and wait on per-tile-row counters
    (larva_conv3x3_pipeline_*; csrc/conv3x3_pipe.inc).  Every layer's output equals the per-layer launch's bit for bit --
    M4B4 on a 339 x 510 image (33 layers), the V2 shape with its 192 -> 48 merge conv, two images, and a tensor of fewer
    tiles than workgroup slots -- three times over (stale bytes from a private L2 would differ from run to run).
    A measurement-library kernel since its timing came out behind the per-layer launches (profiles/r05_layer_pipeline.txt)."""
    from larvanet_amd import kernels as K
    D = _diag_lib_or_skip()
    P = (W + 3) // 4 * 4
    rng = np.random.default_rng(N * 131 + H + W + bpm)

    def padded(shape, scale):
        t = torch.zeros(shape[:-1] + (P,), device=hip_device)
        t[..., :W] = _dev(_rand(rng, shape, scale), hip_device)
        return t

    x = padded((N, 48, H, W), 1.0)
    specs = []   # (src indices (-1 = x), relu, res0 index, res1 index) -- indices into the layer list
    cur = -1
    mods = []
    for m in range(modules):
        m_in = cur
        for j in range(bpm):
            b_in = cur
            specs.append(([cur], True, None, None))
            h = len(specs) - 1
            specs.append(([h], False, b_in, m_in if j == bpm - 1 else None))
            cur = len(specs) - 1
        mods.append(cur)
    if merge:
        specs.append((list(mods), False, None, None))
        cur = len(specs) - 1
    specs.append(([cur], True, None, None))
    weights = []
    for srcs, relu, r0, r1 in specs:
        cin = 48 * len(srcs)
        fwd, _ = K.pack_weights(_dev(_rand(rng, (48, cin, 3, 3), (1.0 / (9 * cin)) ** 0.5), hip_device))
        weights.append((fwd, _dev(_rand(rng, (48,), 0.1), hip_device)))
    # one launch per layer
    ref = []
    t = lambda i: x if i == -1 else ref[i]
    for (srcs, relu, r0, r1), (fwd, b) in zip(specs, weights):
        ref.append(K.conv3x3([t(i) for i in srcs], fwd, 48, bias=b, relu=relu, res0=None if r0 is None else t(r0),
                             res1=None if r1 is None else t(r1), logical_w=W))
    # the pipeline on buffers of its own
    outs = [torch.empty(N, 48, H, P, device=hip_device) for _ in specs]
    u = lambda i: x if i == -1 else outs[i]
    layers = [dict(srcs=[u(i) for i in srcs], wpk=fwd, bias=b, relu=relu, res0=None if r0 is None else u(r0),
                   res1=None if r1 is None else u(r1), out=outs[k], dep=max(srcs))
              for k, ((srcs, relu, r0, r1), (fwd, b)) in enumerate(zip(specs, weights))]
    pipe = D.ConvPipeline(layers, logical_w=W)
    assert pipe.supported
    for rep in range(3):
        for o in outs:
            o.fill_(float("nan"))
        pipe.run()
        torch.cuda.synchronize()
        pipe.check()
        for k, (o, r) in enumerate(zip(outs, ref)):
            assert torch.equal(o, r), "layer %d of %d differs (run %d): max |d| %g" % (k, len(specs), rep, float((o - r).abs().nan_to_num(1e30).max()))
    assert float(ref[-1].abs().max()) > 1e-3


def test_layer_pipeline_refuses_what_it_cannot_order(hip_device):
    """larva_conv3x3_pipeline_plan: a layer that reads a tensor written by a layer its `dep` does not wait for, two layers
    writing one tensor, and a dep that is not an earlier layer are refused (hipErrorInvalidValue)."""
    from larvanet_amd import kernels as K
    D = _diag_lib_or_skip()
    x = torch.zeros(1, 48, 30, 48, device=hip_device)
    fwd, _ = K.pack_weights(torch.zeros(48, 48, 3, 3, device=hip_device))
    a, b, c = (torch.empty_like(x) for _ in range(3))
    ok = [dict(srcs=[x], wpk=fwd, out=a, dep=-1), dict(srcs=[a], wpk=fwd, out=b, dep=0), dict(srcs=[b], wpk=fwd, res0=a, out=c, dep=1)]
    assert D.ConvPipeline(ok).supported
    for bad in ([ok[0], dict(srcs=[a], wpk=fwd, out=b, dep=-1)],                       # reads layer 0's output without waiting for it
                [ok[0], dict(srcs=[a], wpk=fwd, out=a, dep=0)],                        # two layers write one tensor
                [ok[0], dict(srcs=[x], wpk=fwd, out=b, dep=-1), dict(srcs=[b], wpk=fwd, res0=a, out=c, dep=1)],   # res0 not behind dep
                [dict(srcs=[x], wpk=fwd, out=a, dep=0)]):                              # dep is not an earlier layer
        with pytest.raises(RuntimeError, match="hip error 1"):
            D.ConvPipeline(bad)


def test_pair_chain_probe_equals_the_strip_launches_bit_for_bit(hip_device):
    """The round-5 one-launch chain (both half-batch chains in one launch, LDS flags inside a workgroup, per-tile inboxes
    between workgroups: csrc/conv3x3_pair_chain.inc, measurement library only) run through its own tool on a short chain:
    identical to the 2 x layers strip launches bit for bit, no bounded wait expired, with and without the phase lock."""
    import os
    import subprocess
    import sys
    _diag_lib_or_skip()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PAIR_LOCK="0,3")
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "probe_pair_chain.py"), "6"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("one launch vs")]
    assert len(lines) == 2, res.stdout[-2000:]
    for l in lines:
        assert "bit-identical True" in l and "gave up waiting: 0" in l, l


def test_timed_launches_compute_what_the_plain_launches_compute(hip_device):
    """larva_conv3x3_fwd_timed / larva_conv3x3_fwd_strips_timed (kernel-attached event timings: bench.py's
    `launch_alone_ms`): entry points of the MEASUREMENT library (tools/build_diag.sh, tools/larva_diag.h), not of the
    product ABI.  Same output as the product's untimed launch, durations positive and ordered (min <= mean)."""
    import importlib.util
    import os
    from larvanet_amd import kernels as K
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("diag_lib", os.path.join(root, "tools", "diag_lib.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    if not D.available():
        pytest.skip("tools/_diag/diag.so has not been built (tools/build_diag.sh diag)")
    gen = torch.Generator().manual_seed(8)
    x = (torch.randn(4, 48, 48, 48, generator=gen) * 20).to(hip_device)
    w = (torch.randn(48, 48, 3, 3, generator=gen) * 0.05).to(hip_device)
    b = torch.randn(48, generator=gen).to(hip_device)
    fwd, _ = K.pack_weights(w)
    ref = K.conv3x3(x, fwd, 48, bias=b, relu=True)
    out = torch.full_like(ref, float("nan"))
    mean, best = D.conv3x3_relu_timed(x, fwd, 48, b, out, 3)
    assert torch.equal(out, ref) and 0 < best <= mean < 1.0
    out = torch.full_like(ref, float("nan"))
    mean, best = D.conv3x3_strips_timed(x, fwd, 48, b, out, 3, images=(1, 3), relu=True)
    torch.cuda.synchronize()
    assert torch.equal(out[1:3], ref[1:3]) and bool(torch.isnan(out[0]).all()) and bool(torch.isnan(out[3]).all())
    assert 0 < best <= mean < 1.0
