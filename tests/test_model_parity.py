"""Whole-network parity of the plugin (larvanet_amd.models.LarvaNet) against the vectors captured
from the imported reference, and against the torch CPU restatement at BASELINE size.  Tolerance
for fp32 forward outputs on the 0-255 scale: 2e-3 absolute (north_star: within 1e-3 dB PSNR);
measured differences are ~1e-4."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(name, argv, training=False, seed=0):
    import importlib
    mod = importlib.import_module("larvanet_amd.models." + name)
    m = mod.create_model()
    m.parse_args(argv)
    torch.manual_seed(seed)
    m.prepare(is_training=training, scales=[4])
    return m


def _load_sd(m, npz, prefix="sd."):
    sd = {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}
    return sd


class FakeValLoader:
    def __init__(self, seed):
        rng = np.random.RandomState(seed)
        self.pairs = []
        for (h, w) in ((10, 12), (9, 14)):
            lr = rng.randint(0, 256, size=(3, h, w)).astype(np.float32)
            hr = rng.randint(0, 256, size=(3, 4 * h + 1, 4 * w + 2)).astype(np.float32)
            self.pairs.append((lr, hr))

    def get_num_images(self):
        return len(self.pairs)

    def get_image_pair(self, image_index, scale):
        lr, hr = self.pairs[image_index]
        return lr, hr, "img%d" % image_index


def test_staged_forward_f1(hip_device, golden):
    g = golden("f1_m2b2_forward.npz")
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"])
    sd = m.model.state_dict()
    for k in sd:  # same seed, same draw order as the reference -> identical initial weights
        assert np.array_equal(sd[k].cpu().numpy(), g["sd." + k]), k
    net = m.model
    x = torch.from_numpy(g["x"]).to(hip_device)
    with torch.no_grad():
        fea = net.head(x)
        np.testing.assert_allclose(fea.cpu().numpy(), g["stage.head"], rtol=0, atol=5e-4)
        base = net.base(x)
        np.testing.assert_allclose(base.cpu().numpy(), g["base"], rtol=0, atol=5e-4)
        for i in range(2):
            fea = getattr(net, "body_%d" % i)(fea)
            np.testing.assert_allclose(fea.cpu().numpy(), g["stage.body_%d" % i], rtol=0, atol=1e-3)
            out = getattr(net, "body_%d" % i).leg(fea, base)
            np.testing.assert_allclose(out.cpu().numpy(), g["exit_%d" % i], rtol=0, atol=2e-3)
        np.testing.assert_allclose(net(x).cpu().numpy(), g["final"], rtol=0, atol=2e-3)


def test_train_step_larva_f5(hip_device, golden):
    """The reference's own train_step_larva on the same weights/batch: loss of 3 steps, every
    parameter gradient of step 1, weights after 3 AdamW steps, bookkeeping."""
    g = golden("f5_train_steps.npz")
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], training=True)
    m.volume_per_step = 12 * 12 * 2 * 3
    x = torch.from_numpy(g["x"]).to(hip_device)
    truth = torch.from_numpy(g["truth"]).to(hip_device)
    args = types.SimpleNamespace(train_path="/tmp")
    val = FakeValLoader(7)
    losses = []
    for step in range(3):
        losses.append(m.train_step_larva(args, val, x, truth, None))
        if step == 0:
            for k, p in m.model.named_parameters():
                ref = g["grad1." + k]
                got = p.grad.cpu().numpy()
                tol = 2e-4 * max(float(np.abs(ref).max()), 1e-6)
                assert float(np.abs(got - ref).max()) <= tol, (k, float(np.abs(got - ref).max()), tol)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    after = {k: v.cpu().numpy() for k, v in m.model.state_dict().items()}
    flat = np.concatenate([after[k].ravel() for k in sorted(after)])
    np.testing.assert_allclose(flat[::61], g["after3_sample"], rtol=0, atol=2e-5)
    assert m.global_step == int(g["global_step"]) and m.temp_volume == int(g["temp_volume"])
    np.testing.assert_allclose(m.get_lr(), g["lrs"][-1])


def test_canonical_forward_f6(hip_device, golden):
    g = golden("f6_m4b4_canonical.npz")
    m = _model("LarvaNet", ["--num_modules=4", "--num_blocks=4,4,4,4"])
    sd = m.model.state_dict()
    flat = np.concatenate([sd[k].cpu().numpy().ravel() for k in sorted(sd)])
    np.testing.assert_array_equal(flat[::997], g["sd_sample"])
    x = (torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255).to(hip_device)
    with torch.no_grad():
        y = m.model(x).cpu().numpy()
    assert y.shape == (16, 3, 192, 192)
    d = np.abs(y.ravel()[g["sample_idx"]] - g["sample_val"])
    assert float(d.max()) < 2e-3, float(d.max())
    # (the uint8-protocol image and its PSNR at this size: tests/test_headline_parity.py, fixture F12)


def test_upscale_psnr_f10(hip_device, golden):
    from larvanet_amd import metrics
    g = golden("f10_upscale_psnr.npz")
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"])
    up = m.upscale(input_list=[g["lr"]], scale=4)
    assert up.shape == (1, 3, 80, 104) and up.dtype == np.float32
    np.testing.assert_allclose(up[0], g["up"], rtol=0, atol=2e-3)
    o8 = metrics.image_to_uint8(up[0])
    t8 = metrics.fit_truth_image_size(o8, metrics.image_to_uint8(g["hr"]))
    assert abs(float(metrics.image_psnr(o8, t8)) - float(g["psnr"])) < 1e-3


def test_save_restore_roundtrip(hip_device, tmp_path):
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=1,1"], training=True, seed=3)
    path = m.save(str(tmp_path))
    sd = torch.load(path, map_location="cpu")
    assert sorted(sd) == sorted(m.model.state_dict())
    m2 = _model("LarvaNet", ["--num_modules=2", "--num_blocks=1,1"], seed=4)
    m2.restore(path)
    x = np.random.RandomState(0).randint(0, 256, size=(3, 9, 11)).astype(np.float32)
    assert np.array_equal(m.upscale([x], 4), m2.upscale([x], 4))


def test_hip_graph_replay_matches_eager_launches(hip_device):
    """The captured forward+backward must be the same arithmetic as launching kernel by kernel."""
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(hip_device)
    t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(hip_device)
    args = types.SimpleNamespace(train_path="/tmp")
    results = []
    for use_graph in (False, True):
        m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,1"], training=True, seed=5)
        m.use_hip_graph = use_graph
        losses = [m.train_step_larva(args, FakeValLoader(7), x, t) for _ in range(4)]
        results.append((losses, {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}))
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert np.array_equal(results[0][1][k], results[1][1][k]), k


@pytest.mark.parametrize("name,flags", [("LarvaNet", ["--num_modules=2", "--num_blocks=2,1"]),
                                        ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"])])
def test_deferred_wgrad_trains_like_per_node_wgrad(hip_device, name, flags):
    """Weight gradients issued in a few launches at the end of backward: the split-K grouping
    differs from the per-node launches, so equality holds to rounding, not bit for bit."""
    g = torch.Generator().manual_seed(19)
    x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(hip_device)
    t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(hip_device)
    results = []
    for defer in (False, True):
        m = _model(name, flags, training=True, seed=5)
        m.use_hip_graph = False
        m.defer_wgrad = defer
        loss, _ = m._forward_backward(x, t)
        torch.cuda.synchronize()
        results.append((float(loss.detach()), {k: p.grad.cpu().numpy().copy() for k, p in m.model.named_parameters()}))
    assert results[0][0] == results[1][0]  # the forward pass does not depend on the mode
    for k, ga in results[0][1].items():
        gb = results[1][1][k]
        # 1e-5 of the tensor's largest gradient: fp32 split-K regrouping, nothing more
        assert np.abs(ga - gb).max() <= 1e-5 * max(np.abs(ga).max(), 1e-30), k


@pytest.mark.parametrize("name,flags", [("LarvaNet", ["--num_modules=3", "--num_blocks=2,1,1"]),
                                        ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"])])
def test_joint_input_gradient_launch_matches_separate_launches(hip_device, name, flags):
    """d fea_i as ONE conv over [dh_body ; dh_leg] == two convs + autograd's add, to fp32 rounding
    (the K loop is one chain of 96 channels instead of two chains of 48 added afterwards)."""
    g = torch.Generator().manual_seed(23)
    x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(hip_device)
    t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(hip_device)
    results = []
    for joint in (False, True):
        m = _model(name, flags, training=True, seed=5)
        m.use_hip_graph = False
        m.joint_input_grads = joint
        loss, _ = m._forward_backward(x, t)
        torch.cuda.synchronize()
        results.append((float(loss.detach()), {k: p.grad.cpu().numpy().copy() for k, p in m.model.named_parameters()}))
    assert results[0][0] == results[1][0]
    for k, ga in results[0][1].items():
        gb = results[1][1][k]
        assert np.abs(ga - gb).max() <= 2e-5 * max(np.abs(ga).max(), 1e-30), k


@pytest.mark.parametrize("name,flags", [("LarvaNet", ["--num_modules=3", "--num_blocks=2,1,1"]),
                                        ("LarvaNet", ["--num_modules=5", "--num_blocks=1,1,1,1,1"]),
                                        ("LarvaNet", ["--num_modules=9", "--num_blocks=1,1,1,1,1,1,1,1,1"]),
                                        ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"])])
def test_exits_as_one_batched_node_match_one_node_per_exit(hip_device, name, flags):
    """ExitsFn (all exits after the body chain, batched launches, joint input gradients) against
    the reference's interleaved order with one autograd node per exit: the same last image bit for
    bit (forward arithmetic is identical), the loss to fp32 rounding (the batched exits add up
    |out - truth| per conv workgroup inside the conv launch, the per-exit path per block of a
    separate sweep), gradients to fp32 rounding (the joint dgrad sums in one K chain what the other
    path adds afterwards)."""
    g = torch.Generator().manual_seed(29)
    x = (torch.rand(2, 3, 12, 16, generator=g) * 255).to(hip_device)
    t = (torch.rand(2, 3, 48, 64, generator=g) * 255).to(hip_device)
    results = []
    for batched in (False, True):
        m = _model(name, flags, training=True, seed=5)
        m.use_hip_graph = False
        m.batch_exits = batched
        loss, out = m._forward_backward(x, t)
        torch.cuda.synchronize()
        results.append((float(loss.detach()), out.detach().cpu().numpy().copy(),
                        {k: p.grad.cpu().numpy().copy() for k, p in m.model.named_parameters()}))
    assert abs(results[0][0] - results[1][0]) <= 1e-6 * abs(results[0][0])
    assert np.array_equal(results[0][1], results[1][1])
    for k, ga in results[0][2].items():
        gb = results[1][2][k]
        assert np.abs(ga - gb).max() <= 2e-5 * max(np.abs(ga).max(), 1e-30), k


def test_v2_tail_f8(hip_device, golden):
    """LarvaNetV2: merge conv over the un-materialised concatenation, tail exit, (M+1)-way loss."""
    g = golden("f8_v2_tail.npz")
    m = _model("LarvaNetV2", ["--num_modules=2", "--num_blocks=2,2"], training=True)
    sd = m.model.state_dict()
    for k in sd:  # seed-0 construction order matches the reference
        if k.startswith("tail."):
            assert np.array_equal(sd[k].cpu().numpy(), g["sd." + k]), k
    x = torch.from_numpy(g["x"]).to(hip_device)
    truth = torch.from_numpy(g["truth"]).to(hip_device)
    with torch.no_grad():
        out = m.model(x).cpu().numpy()
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=2e-3)
    m.use_hip_graph = False
    m.volume_per_step = 0
    loss = m.train_step_larva(types.SimpleNamespace(train_path="/tmp"), FakeValLoader(7), x, truth)
    assert abs(loss - float(g["loss"])) < 2e-5 * float(g["loss"])
    grads = {k: p.grad.cpu().numpy() for k, p in m.model.named_parameters()}
    for k in grads:
        if k.startswith("tail."):
            ref = g["grad." + k]
            tol = 2e-4 * max(float(np.abs(ref).max()), 1e-6)
            assert float(np.abs(grads[k] - ref).max()) <= tol, k
    for k, ref_abs in zip(g["trunk_keys"], g["trunk_gabs"]):
        got = float(np.abs(grads[str(k)].astype(np.float64)).sum())
        assert abs(got - float(ref_abs)) <= 1e-3 * float(ref_abs) + 1e-7, (str(k), got, float(ref_abs))


def test_v2_restore_takes_matching_keys_only(hip_device, tmp_path):
    v1 = _model("LarvaNet", ["--num_modules=2", "--num_blocks=1,1"], training=True, seed=3)
    path = v1.save(str(tmp_path))
    v2 = _model("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,1"], seed=4)
    tail_before = v2.model.tail.merge_conv.weight.detach().cpu().clone()
    v2.restore(path)
    assert torch.equal(v2.model.head.feature_extraction.weight.cpu(), v1.model.head.feature_extraction.weight.cpu())
    assert torch.equal(v2.model.tail.merge_conv.weight.cpu(), tail_before)


@pytest.mark.parametrize("plugin", ["LarvaLeg", "LarvaLegV2"])
def test_early_exit_plugin_matches_staged_exits(hip_device, golden, plugin):
    """models/LarvaLeg.py:289-300 / models/LarvaLegV2.py:358-367: --leg=k returns exit k; k = 0 the bicubic base alone.
    (LarvaLegV2 = the V2 network, tail.* parameters included but not evaluated; its trunk draws the same initial
    weights as V1's under the same seed, so F1's staged exits are its exits too.)"""
    g = golden("f1_m2b2_forward.npz")
    x = [g["x"][0], g["x"][1]]
    for leg, ref in ((0, g["base"]), (1, g["exit_0"]), (2, g["exit_1"])):
        m = _model(plugin, ["--num_modules=2", "--num_blocks=2,2", "--leg=%d" % leg])
        assert ("tail.merge_conv.weight" in m.model.state_dict()) == (plugin == "LarvaLegV2")
        out = m.upscale(x, 4)
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-3)


def test_device_resident_training_and_runtime_probe(hip_device, tmp_path, capsys):
    from larvanet_amd import runtime, train_larva
    model = train_larva.main([
        "--model=LarvaNet", "--dataloader=device_patch_loader", "--device_source=synthetic_loader",
        "--val_dataloader=synthetic_loader", "--train_path", str(tmp_path), "--max_steps=3", "--batch_size=4",
        "--input_patch_size=12", "--num_modules=1", "--num_blocks=1", "--synthetic_images=3",
        "--synthetic_lr_size=20", "--data_seed=1"])
    assert model.global_step == 3
    res = runtime.main(["--model=LarvaNet", "--dataloader=synthetic_loader", "--num_modules=1", "--num_blocks=1",
                        "--synthetic_images=2", "--synthetic_lr_size=20"])
    assert res[4] > 0


def test_step_reads_batches_produced_into_its_input_buffers(hip_device):
    """A producer that fills input_buffers() in place trains exactly like one that hands over fresh
    tensors (which the step copies into the captured graph's inputs)."""
    g = torch.Generator().manual_seed(41)
    batches = [((torch.rand(2, 3, 12, 16, generator=g) * 255).to(hip_device),
                (torch.rand(2, 3, 48, 64, generator=g) * 255).to(hip_device)) for _ in range(4)]
    args = types.SimpleNamespace(train_path="/tmp")
    finals = []
    for in_place in (False, True):
        m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=1,1"], training=True, seed=5)
        assert m.input_buffers((2, 3, 12, 16), (2, 3, 48, 64)) is None  # nothing captured yet
        losses = []
        for x, t in batches:
            bufs = m.input_buffers(x.shape, t.shape) if in_place else None
            if bufs is not None:
                bufs[0].copy_(x)
                bufs[1].copy_(t)
                x, t = bufs
            losses.append(m.train_step_larva(args, FakeValLoader(7), x, t))
        assert not in_place or m.input_buffers((2, 3, 12, 16), (2, 3, 48, 64)) is not None
        finals.append((losses, {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}))
    assert finals[0][0] == finals[1][0]
    for k in finals[0][1]:
        assert np.array_equal(finals[0][1][k], finals[1][1][k]), k


def test_training_state_resume_is_bit_exact(hip_device, tmp_path):
    """weights + optimizer moments + scheduler + counters: 2 steps, save, 2 more == 4 straight."""
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(2, 3, 12, 12, generator=g) * 255).to(hip_device)
    t = (torch.rand(2, 3, 48, 48, generator=g) * 255).to(hip_device)
    args = types.SimpleNamespace(train_path=str(tmp_path))
    val = FakeValLoader(7)
    a = _model("LarvaNet", ["--num_modules=1", "--num_blocks=2"], training=True, seed=2)
    la = [a.train_step_larva(args, val, x, t) for _ in range(4)]
    b = _model("LarvaNet", ["--num_modules=1", "--num_blocks=2"], training=True, seed=2)
    lb = [b.train_step_larva(args, val, x, t) for _ in range(2)]
    wpath, spath = b.save(str(tmp_path)), b.save_training_state(str(tmp_path))
    c = _model("LarvaNet", ["--num_modules=1", "--num_blocks=2"], training=True, seed=99)
    c.restore(wpath)
    c.restore_training_state(spath)
    assert c.global_step == 2
    lc = [c.train_step_larva(args, val, x, t) for _ in range(2)]
    assert lb + lc == la
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, c.model.state_dict()[k]), k


@pytest.mark.parametrize("name,flags,halo", [("LarvaNet", ["--num_modules=2", "--num_blocks=2,1"], 9),
                                             ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"], 10),
                                             ("LarvaLeg", ["--num_modules=2", "--num_blocks=2,1", "--leg=1"], 7)])
def test_row_bands_with_receptive_halo_reproduce_the_full_image(hip_device, name, flags, halo):
    """one image cut into row bands (one per GPU), each computed from its rows
    plus the receptive halo, is the full-image result bit for bit; one row less of halo is not."""
    from larvanet_amd import image_utils
    m = _model(name, flags, seed=3)
    assert m.receptive_halo() == halo
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=(3, 61, 37)).astype(np.float32)
    full = m.upscale([img], 4)[0]
    world = 4
    bands = [image_utils.upscale_banded(m, img, 4, r, world, lambda mine, r=r: [
        image_utils.upscale_band(m, img, 4, *image_utils.band_rows(img.shape[1], world)[q], halo) for q in range(world)])
        for r in range(1)]
    assert bands[0].shape == full.shape and np.array_equal(bands[0], full)
    short = np.concatenate([image_utils.upscale_band(m, img, 4, r0, r1, halo - 1)
                            for r0, r1 in image_utils.band_rows(img.shape[1], world)], axis=1)
    assert not np.array_equal(short, full)  # the halo is tight
    assert image_utils.band_rows(5, 8)[0] == (0, 0) and image_utils.band_rows(5, 8)[-1] == (4, 5)


def test_forward_and_train_step_shape_fuzz_against_the_cpu_restatement(hip_device):
    """Seeded random network configurations (1..4 bodies of 1..3 blocks, V1 and V2) and image
    sizes (aligned and odd): inference forward and one training step's loss and gradients against
    oracle/larva_torch.py (the reference's own torch operators on the CPU) with the same weights."""
    from oracle import larva_torch as T
    rng = np.random.RandomState(4242)
    for case in range(6):
        M = int(rng.randint(1, 5))
        blocks = [int(rng.randint(1, 4)) for _ in range(M)]
        v2 = case % 3 == 2
        name = "LarvaNetV2" if v2 else "LarvaNet"
        flags = ["--num_modules=%d" % M, "--num_blocks=%s" % ",".join(map(str, blocks))]
        m = _model(name, flags, training=True, seed=100 + case)
        sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
        # inference on an odd-sized image
        h, w = int(rng.randint(5, 30)), int(rng.randint(5, 40))
        img = rng.randint(0, 256, size=(3, h, w)).astype(np.float32)
        got = m.upscale([img], 4)[0]
        with torch.no_grad():
            xt = torch.from_numpy(img)[None]
            ref = (T.forward_v2(sd, xt, blocks) if v2 else T.forward(sd, xt, blocks))[0].numpy()
        assert np.abs(got - ref).max() <= 2e-3, (case, name, blocks, h, w, float(np.abs(got - ref).max()))
        # one training step: loss and gradients
        n, p = int(rng.randint(1, 4)), 4 * int(rng.randint(2, 5))
        x = torch.from_numpy(rng.randint(0, 256, size=(n, 3, p, p)).astype(np.float32))
        t = torch.from_numpy(rng.randint(0, 256, size=(n, 3, 4 * p, 4 * p)).astype(np.float32))
        m.use_hip_graph = False
        loss, _ = m._forward_backward(x.to(hip_device), t.to(hip_device))
        torch.cuda.synchronize()
        sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref_loss = T.multi_exit_loss(sd_req, x, t, blocks, v2=v2)
        ref_loss.backward()
        ref_value = float(ref_loss.detach())
        assert abs(float(loss.detach()) - ref_value) <= 2e-5 * abs(ref_value), (case, float(loss.detach()), ref_value)
        for k, prm in m.model.named_parameters():
            ga, gb = prm.grad.cpu().numpy(), sd_req[k].grad.numpy()
            assert np.abs(ga - gb).max() <= 2e-4 * max(np.abs(gb).max(), 1e-30), (case, name, blocks, k)


def test_train_step_on_a_patch_larger_than_the_direct_head_threshold(hip_device):
    """LarvaHead picks its direct K = 27 kernel for INFERENCE on more than 100 k LR pixels; a training step on that many
    pixels (2 x 3 x 232 x 232) must keep the padded-MFMA launch and the padded input its weight gradient reads (inside an
    autograd.Function the grad mode is always off: the flag comes from the module).  Loss and every gradient against
    oracle/larva_torch.py; the same image through upscale() afterwards takes the direct kernel and matches too."""
    from oracle import larva_torch as T
    from larvanet_amd.autograd import is_large_inference
    blocks = [1]
    m = _model("LarvaNet", ["--num_modules=1", "--num_blocks=1"], training=True, seed=7)
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    rng = np.random.RandomState(77)
    n, p = 2, 232
    assert is_large_inference(n, p, p)
    x = torch.from_numpy(rng.randint(0, 256, size=(n, 3, p, p)).astype(np.float32))
    t = torch.from_numpy(rng.randint(0, 256, size=(n, 3, 4 * p, 4 * p)).astype(np.float32))
    m.use_hip_graph = False
    loss, _ = m._forward_backward(x.to(hip_device), t.to(hip_device))
    torch.cuda.synchronize()
    sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss = T.multi_exit_loss(sd_req, x, t, blocks)
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach()))
    for k, prm in m.model.named_parameters():
        ga, gb = prm.grad.cpu().numpy(), sd_req[k].grad.numpy()
        assert np.abs(ga - gb).max() <= 2e-4 * max(np.abs(gb).max(), 1e-30), k
    got = m.upscale([x[0].numpy(), x[1].numpy()], 4)
    with torch.no_grad():
        ref = T.forward(sd, x, blocks).numpy()
    assert np.abs(got - ref).max() <= 2e-3


@pytest.mark.parametrize("name,flags", [("LarvaNet", ["--num_modules=3", "--num_blocks=2,1,2"]),
                                        ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"])])
def test_early_loss_capture_trains_exactly_like_the_single_graph(hip_device, name, flags):
    """`return loss.item()` (models/LarvaNet.py:139) with a captured step, the host waiting for the forward only --
    "poll": the launch that finishes the loss stores it into coherent pinned host memory, which the host polls;
    "split": forward | backward as two graphs, the loss copied out between them -- must return the same floats and
    leave the same weights as one graph with the loss read at the end, and as the device-tensor return."""
    g = torch.Generator().manual_seed(19)
    xs = [(torch.rand(4, 3, 12, 16, generator=g) * 255).to(hip_device) for _ in range(5)]
    ts = [(torch.rand(4, 3, 48, 64, generator=g) * 255).to(hip_device) for _ in range(5)]
    args = types.SimpleNamespace(train_path="/tmp")
    results = []
    for early, sync in (("poll", True), ("split", True), (False, True), ("poll", False)):
        m = _model(name, flags, training=True, seed=5)
        m.early_loss, m.sync_loss = early, sync
        losses = [m.train_step_larva(args, FakeValLoader(7), x, t) for x, t in zip(xs, ts)]
        assert (m._graph_back is not None) == (early == "split" and sync) and m._graph_polls == (early == "poll" and sync)
        if sync:
            assert all(isinstance(v, float) for v in losses)
        else:
            losses = [float(v) for v in losses]
        results.append((losses, {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}))
    for other in results[1:]:
        assert results[0][0] == other[0]
        for k in results[0][1]:
            assert np.array_equal(results[0][1][k], other[1][k]), k


@pytest.mark.parametrize("name", ["LarvaNet", "LarvaNetV2"])
@pytest.mark.parametrize("mode", ["bilinear"])
def test_other_interpolate_modes_against_the_cpu_restatement(hip_device, name, mode):
    """--interpolate=<mode> (models/LarvaNet.py:57,283-285): inference forward and one training step (loss and
    every gradient: the base image carries no gradient but shifts every exit's L1 sign) against
    oracle/larva_torch.py, which hands the same string to F.interpolate."""
    from oracle import larva_torch as T
    blocks, v2 = [2, 1], name == "LarvaNetV2"
    m = _model(name, ["--num_modules=2", "--num_blocks=2,1", "--interpolate=" + mode], training=True, seed=7)
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    rng = np.random.RandomState(77)
    img = rng.randint(0, 256, size=(3, 13, 18)).astype(np.float32)
    got = m.upscale([img], 4)[0]
    with torch.no_grad():
        xt = torch.from_numpy(img)[None]
        ref = (T.forward_v2(sd, xt, blocks, mode) if v2 else T.forward(sd, xt, blocks, mode))[0].numpy()
    assert np.abs(got - ref).max() <= 2e-3, (name, mode, float(np.abs(got - ref).max()))
    x = torch.from_numpy(rng.randint(0, 256, size=(2, 3, 12, 12)).astype(np.float32))
    t = torch.from_numpy(rng.randint(0, 256, size=(2, 3, 48, 48)).astype(np.float32))
    for use_graph in (False, True):
        m.use_hip_graph = use_graph
        m._graph_shape = None
        loss, _ = m._forward_backward(x.to(hip_device), t.to(hip_device))
        torch.cuda.synchronize()
        sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref_loss = T.multi_exit_loss(sd_req, x, t, blocks, v2=v2, mode=mode)
        ref_loss.backward()
        assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach())), (name, mode, use_graph)
        for k, prm in m.model.named_parameters():
            ga, gb = prm.grad.cpu().numpy(), sd_req[k].grad.numpy()
            assert np.abs(ga - gb).max() <= 2e-4 * max(np.abs(gb).max(), 1e-30), (name, mode, use_graph, k)


@pytest.mark.parametrize("name,flags,use_graph", [("LarvaNet", ["--num_modules=3", "--num_blocks=2,1,2"], False),
                                                  ("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], True),
                                                  ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"], False),
                                                  ("LarvaNetV2", ["--num_modules=2", "--num_blocks=2,1"], True)])
def test_two_half_batch_chains_train_exactly_like_one_chain(hip_device, name, flags, use_graph):
    """autograd.DualChain: the body chain as two half-batch chains of strip-tile launches on two
    streams is the same arithmetic as one chain of full-batch launches -- losses and weights after
    three steps are identical bit for bit (eager launches and hipGraph replay; V1 joins lazily, V2
    after every body in backward).  Batch of 5: the halves are 2 and 3 images."""
    g = torch.Generator().manual_seed(37)
    x = (torch.rand(5, 3, 16, 20, generator=g) * 255).to(hip_device)
    t = (torch.rand(5, 3, 64, 80, generator=g) * 255).to(hip_device)
    args = types.SimpleNamespace(train_path="/tmp")
    from larvanet_amd.autograd import DualChain
    results = []
    for dual in (False, True):
        m = _model(name, flags, training=True, seed=5)
        m.use_hip_graph = use_graph
        m.dual_chain = dual
        losses = [m.train_step_larva(args, FakeValLoader(7), x, t) for _ in range(3)]
        assert m.use_hip_graph == use_graph
        assert not DualChain._forked and not DualChain._keep
        results.append((losses, {k: v.cpu().numpy().copy() for k, v in m.model.state_dict().items()}))
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert np.array_equal(results[0][1][k], results[1][1][k]), k


@pytest.mark.parametrize("name,flags", [("LarvaNet", ["--num_modules=2", "--num_blocks=2,1"]),
                                        ("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,2"])])
def test_repeated_inference_shape_replays_a_graph_with_the_eager_result(hip_device, name, flags, tmp_path):
    """A batch shape seen twice is captured (two half-batch chains inside) and replayed: same bits as
    the eager forward, also after the weights change under the captured graph (restore / repack)."""
    m = _model(name, flags, seed=3)
    rng = np.random.RandomState(5)
    batch = [rng.randint(0, 256, size=(3, 16, 20)).astype(np.float32) for _ in range(4)]
    eager = m.upscale(batch, 4)                       # first sight: eager
    assert not getattr(m, "_infer_graphs", None)
    second = m.upscale(batch, 4)                      # second sight: captured + replayed
    assert len(m._infer_graphs) == 1 and all(v is not False for v in m._infer_graphs.values())
    third = m.upscale(batch, 4)
    assert np.array_equal(eager, second) and np.array_equal(eager, third)
    other = m.upscale([b[:, :9, :13] for b in batch], 4)   # another shape: eager, cache untouched
    assert other.shape == (4, 3, 36, 52) and len(m._infer_graphs) == 1
    m2 = _model(name, flags, training=True, seed=11)
    path = m2.save(str(tmp_path))
    m.restore(path)
    replayed = m.upscale(batch, 4)
    fresh = _model(name, flags, seed=99)
    fresh.restore(path)
    assert np.array_equal(replayed, fresh.upscale(batch, 4)) and not np.array_equal(replayed, eager)


def test_reference_checkpoint_file_restores_and_reproduces_its_outputs(hip_device, golden, tmp_path):
    """A checkpoint as the reference writes it (torch.save of the bare state_dict, models/LarvaNet.py:183-185;
    here rebuilt from the weights captured in fixture F1) loads through restore() into a differently
    initialised plugin, which then reproduces the reference's own staged outputs."""
    g = golden("f1_m2b2_forward.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    path = str(tmp_path / "model_step7_vol0G.pth")
    torch.save(sd, path)
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], seed=1234)     # other weights than the file's
    assert not np.array_equal(m.model.state_dict()["head.feature_extraction.weight"].cpu().numpy(),
                              g["sd.head.feature_extraction.weight"])
    m.restore(path)
    up = m.upscale([g["x"][0], g["x"][1]], 4)
    np.testing.assert_allclose(up, g["final"], rtol=0, atol=2e-3)
    # and the other way round: what save() writes is that same wire format
    out = m.save(str(tmp_path))
    back = torch.load(out, map_location="cpu")
    assert sorted(back) == sorted(sd) and all(torch.equal(back[k], sd[k]) for k in sd)


@pytest.mark.parametrize("name,nf,use_graph", [("LarvaNet", 32, True), ("LarvaNet", 64, True), ("LarvaNetV2", 32, False),
                                               ("LarvaNetV2", 64, True), ("LarvaNet", 32, False)])
def test_num_filters_extension_against_the_cpu_restatement(hip_device, name, nf, use_graph):
    """--num_filters=32 / 64 (BASELINE configs 2 / 5 name 32- and 64-channel bodies; SURVEY 8a N1: a build-side
    extension the reference cannot express -- its width is a constant 48 -- with every leg's last conv kept at 48
    outputs).  No reference fixture can exist; the check is against oracle/larva_torch.py built at the same width
    (same initial weights from the same seed, bit for bit): inference forward, two train_step_larva steps' losses,
    EVERY gradient element of the first step, the weights after the second AdamW step."""
    from oracle import larva_torch as T
    v2 = name == "LarvaNetV2"
    blocks = [2, 1]
    m = _model(name, ["--num_modules=2", "--num_blocks=2,1", "--num_filters=%d" % nf], training=True, seed=11)
    m.use_hip_graph = use_graph
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    ref_sd = T.init_state_dict(blocks, v2=v2, seed=11, nf=nf)
    assert sorted(sd) == sorted(ref_sd) and all(torch.equal(sd[k], ref_sd[k]) for k in sd)
    assert tuple(sd["body_1.leg.recon_block.2.weight"].shape) == (48, nf, 3, 3)
    assert tuple(sd["head.feature_extraction.weight"].shape) == (nf, 3, 3, 3)
    g = torch.Generator().manual_seed(nf)
    x = torch.rand(4, 3, 12, 16, generator=g) * 255
    t = torch.rand(4, 3, 48, 64, generator=g) * 255
    with torch.no_grad():
        y = m.model(x.to(hip_device)).cpu()
        y_ref = (T.forward_v2 if v2 else T.forward)(sd, x, blocks)
    assert float((y - y_ref).abs().max()) < 2e-3
    lr = m.get_lr()
    _, ref_grads = T.train_steps(dict(sd), x, t, blocks, steps=1, lr=lr, v2=v2)
    ref_losses, _ = T.train_steps(sd, x, t, blocks, steps=2, lr=lr, v2=v2)     # (sd: the weights after two steps)
    args = types.SimpleNamespace(train_path="/tmp")
    xd, td = x.to(hip_device), t.to(hip_device)
    m.volume_per_step = 12 * 16 * 4 * 3
    losses = []
    for step in range(2):
        losses.append(m.train_step_larva(args, FakeValLoader(7), xd, td))
        if step == 0:
            for k, p in m.model.named_parameters():
                got, ref = p.grad.detach().cpu().numpy(), ref_grads[k].numpy()
                assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-30), k
    assert m.use_hip_graph == use_graph
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for k, v in m.model.state_dict().items():
        d = np.abs(v.cpu().numpy() - sd[k].numpy())
        assert float((d > 2e-5).mean()) < 2e-3 and float(d.max()) <= 2.1 * lr, (k, float(d.max()))
