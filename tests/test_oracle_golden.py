"""The oracle (oracle/) against the vectors captured from the imported reference (tests/golden/).
CPU only.  This is what pins the oracle; the GPU tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from oracle import larva_ref as R
from oracle import larva_torch as T


def _sd(npz, prefix="sd."):
    return {k[len(prefix):]: npz[k] for k in npz.files if k.startswith(prefix)}


def _tsd(npz, prefix="sd."):
    return {k: torch.from_numpy(v) for k, v in _sd(npz, prefix).items()}


def test_c_restatement_stages_f1(golden):
    g = golden("f1_m2b2_forward.npz")
    sd, x = _sd(g), g["x"]
    head = R.head(sd, x)
    np.testing.assert_allclose(head, g["stage.head"], rtol=1e-5, atol=2e-4)
    h = R.relu(R.conv3x3(g["stage.head"], sd["body_0.res_blocks.0.body.0.weight"], sd["body_0.res_blocks.0.body.0.bias"]))
    np.testing.assert_allclose(h, g["stage.body_0.res_blocks.0.relu"], rtol=1e-5, atol=2e-4)
    blk = R.residual_block(sd, "body_0.res_blocks.0", g["stage.head"])
    np.testing.assert_allclose(blk, g["stage.body_0.res_blocks.0"], rtol=1e-5, atol=2e-4)
    b0 = R.body(sd, 0, g["stage.head"], 2)
    np.testing.assert_allclose(b0, g["stage.body_0"], rtol=1e-5, atol=5e-4)
    b1 = R.body(sd, 1, g["stage.body_0"], 2)
    np.testing.assert_allclose(b1, g["stage.body_1"], rtol=1e-5, atol=5e-4)
    base = R.bicubic_up(x, 4)
    np.testing.assert_allclose(base, g["base"], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(R.leg(sd, "body_0.leg", g["stage.body_0"], g["base"]), g["exit_0"], rtol=1e-5, atol=5e-4)
    np.testing.assert_allclose(R.forward(sd, x, [2, 2]), g["final"], rtol=1e-5, atol=2e-3)
    assert np.array_equal(g["exit_1"], g["final"])  # forward() == last exit (models/LarvaNet.py:287-293)


def test_torch_restatement_f1_and_init(golden):
    g = golden("f1_m2b2_forward.npz")
    sd, x = _tsd(g), torch.from_numpy(g["x"])
    with torch.no_grad():
        outs, feats, base = T.forward_exits(sd, x, [2, 2])
        np.testing.assert_allclose(outs[0].numpy(), g["exit_0"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(outs[1].numpy(), g["exit_1"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(feats[1].numpy(), g["stage.body_1"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(T.forward(sd, x, [2, 2]).numpy(), g["final"], rtol=0, atol=1e-4)
    init = T.init_state_dict([2, 2], seed=0)
    assert sorted(init) == sorted(sd)
    for k in sd:
        assert torch.equal(init[k], sd[k]), k  # same draw order as the reference modules


def test_pixel_shuffle_bit_exact_f3(golden):
    g = golden("f3_pixel_shuffle.npz")
    out = R.pixel_shuffle(g["inp"], 4)
    assert out.dtype == np.int32 and np.array_equal(out, g["out"])
    assert np.array_equal(R.pixel_unshuffle(g["out"], 4), g["inp"])


def test_bicubic_f4(golden):
    g = golden("f4_bicubic.npz")
    np.testing.assert_allclose(R.bicubic_up(g["inp"], 4), g["out"], rtol=1e-5, atol=2e-4)
    # closed-form phase weights quoted in SURVEY 8(a7)
    one = np.zeros((1, 1, 9, 9), np.float32)
    one[0, 0, 4, 4] = 1.0
    up = R.bicubic_up(one, 4)
    col = up[0, 0, 16, 4 * 4 - 8:4 * 4 + 12:4]  # phase 0 row, taps of column phase 0 at successive LR offsets
    assert np.isclose(up[0, 0, 16, 16], 0.74951172 * 0.74951172, atol=1e-6), col


def test_train_steps_f5(golden):
    g = golden("f5_train_steps.npz")
    f1 = golden("f1_m2b2_forward.npz")
    sd = _tsd(f1)  # same seed-0 M2B2 weights
    x, truth = torch.from_numpy(g["x"]), torch.from_numpy(g["truth"])
    # loss by the C restatement (forward only)
    loss_c = R.multi_exit_loss(_sd(f1), g["x"], g["truth"], [2, 2])
    assert abs(loss_c - g["losses"][0]) < 1e-4 * abs(g["losses"][0])
    losses, _ = T.train_steps({k: v.clone() for k, v in sd.items()}, x, truth, [2, 2], steps=1)
    sd3 = {k: v.clone() for k, v in sd.items()}
    losses3, _ = T.train_steps(sd3, x, truth, [2, 2], steps=3)
    np.testing.assert_allclose(losses3, g["losses"], rtol=1e-5)
    # gradients of step 1
    _, grads = T.train_steps({k: v.clone() for k, v in sd.items()}, x, truth, [2, 2], steps=1)
    for k, v in grads.items():
        ref = g["grad1." + k]
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    flat = np.concatenate([sd3[k].numpy().ravel() for k in sorted(sd3)])
    np.testing.assert_allclose(flat[::61], g["after3_sample"], rtol=1e-4, atol=1e-6)
    assert int(g["global_step"]) == 3 and int(g["temp_volume"]) == 3 * 12 * 12 * 2 * 3


def test_c_restatement_grads_f5(golden):
    """dgrad / wgrad / l1-grad of the C restatement chained by hand through the last leg."""
    g = golden("f5_train_steps.npz")
    f1 = golden("f1_m2b2_forward.npz")
    sd = _sd(f1)
    outs, feats, base = R.forward_exits(sd, g["x"], [2, 2])
    M = 2
    # d loss / d exit_1 = sign(out - truth) / (numel * M)
    dout = R.l1_grad(outs[1], g["truth"], 1.0 / M)
    dy2 = R.pixel_unshuffle(dout, 4)
    h = R.relu(R.conv3x3(feats[1], sd["body_1.leg.recon_block.0.weight"], sd["body_1.leg.recon_block.0.bias"]))
    dw2, db2 = R.conv3x3_wgrad(dy2, h)
    np.testing.assert_allclose(dw2, g["grad1.body_1.leg.recon_block.2.weight"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(db2, g["grad1.body_1.leg.recon_block.2.bias"], rtol=2e-4, atol=2e-6)
    dh = R.conv3x3_dgrad(dy2, sd["body_1.leg.recon_block.2.weight"]) * (h > 0)
    dw1, db1 = R.conv3x3_wgrad(dh, feats[1])
    np.testing.assert_allclose(dw1, g["grad1.body_1.leg.recon_block.0.weight"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db1, g["grad1.body_1.leg.recon_block.0.bias"], rtol=2e-4, atol=2e-6)


def test_adamw_restatement():
    rng = np.random.RandomState(0)
    p = rng.randn(1000).astype(np.float32)
    gr = rng.randn(1000).astype(np.float32)
    tp = torch.from_numpy(p.copy()).requires_grad_(True)
    opt = torch.optim.AdamW([tp], lr=4e-4)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    pc = p.copy()
    for step in (1, 2, 3):
        tp.grad = torch.from_numpy(gr * step)
        opt.step()
        pc, m, v = R.adamw(pc, gr * step, m, v, step)
    np.testing.assert_allclose(pc, tp.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_canonical_forward_f6(golden):
    g = golden("f6_m4b4_canonical.npz")
    sd = T.init_state_dict([4, 4, 4, 4], seed=0)
    flat = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    assert flat.size == int(g["n_params"]) == 832704
    np.testing.assert_array_equal(flat[::997], g["sd_sample"])
    x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
    with torch.no_grad():
        y = T.forward(sd, x, [4, 4, 4, 4]).numpy()
    np.testing.assert_allclose(y.ravel()[g["sample_idx"]], g["sample_val"], rtol=0, atol=2e-3)


def test_validate_helpers_f7(golden):
    g = golden("f7_validate_helpers.npz")
    assert np.array_equal(R.image_to_uint8(g["img"]), g["u8"])
    fitted = R.fit_truth_image_size(g["o_img"], g["t_big"])
    assert np.array_equal(fitted, g["fitted"])
    assert abs(R.image_psnr(g["o_img"], g["t_big"]) - float(g["psnr"])) < 1e-4
    from larvanet_amd import metrics
    assert np.array_equal(metrics.image_to_uint8(g["img"]), g["u8"])
    assert abs(float(metrics.image_psnr(g["o_img"], metrics.fit_truth_image_size(g["o_img"], g["t_big"]))) - float(g["psnr"])) < 1e-5


def test_v2_tail_f8(golden):
    g = golden("f8_v2_tail.npz")
    f1 = golden("f1_m2b2_forward.npz")
    assert bool(g["same_trunk"])
    sd = _tsd(f1)
    sd.update(_tsd(g))
    x, truth = torch.from_numpy(g["x"]), torch.from_numpy(g["truth"])
    with torch.no_grad():
        np.testing.assert_allclose(T.forward_v2(sd, x, [2, 2]).numpy(), g["out"], rtol=0, atol=1e-4)
    nsd = {k: v.numpy() for k, v in sd.items()}
    np.testing.assert_allclose(R.forward_v2(nsd, g["x"], [2, 2]), g["out"], rtol=1e-5, atol=2e-3)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = T.multi_exit_loss(params, x, truth, [2, 2], v2=True)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    for k in params:
        if k.startswith("tail."):
            np.testing.assert_allclose(params[k].grad.numpy(), g["grad." + k], rtol=1e-4, atol=1e-5, err_msg=k)
    init = T.init_state_dict([2, 2], v2=True, seed=0)
    for k in sd:
        assert torch.equal(init[k], sd[k]), k


def test_chop_forward_f9(golden):
    g = golden("f9_chop_forward.npz")
    parts = R.split_image(g["img"], 6)
    assert [p.shape for p in parts] == [tuple(s) for s in g["split_shapes"]]
    for i, p in enumerate(parts):
        assert np.array_equal(p, g["split%d" % i])
    ups = [p.repeat(4, axis=1).repeat(4, axis=2) + 0.25 for p in parts]
    assert np.array_equal(R.combine_images(ups, g["img"].shape, 4, 6), g["out"])


def test_upscale_psnr_f10(golden):
    g = golden("f10_upscale_psnr.npz")
    f1 = golden("f1_m2b2_forward.npz")
    sd = _tsd(f1)
    with torch.no_grad():
        up = T.forward(sd, torch.from_numpy(g["lr"])[None], [2, 2])[0].numpy()
    np.testing.assert_allclose(up, g["up"], rtol=0, atol=2e-4)
    o8 = R.image_to_uint8(up)
    assert abs(R.image_psnr(o8, R.image_to_uint8(g["hr"])) - float(g["psnr"])) < 1e-3


def _canonical_batch():
    x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
    truth = torch.rand(16, 3, 192, 192, generator=torch.Generator().manual_seed(1)) * 255
    return x, truth


def test_torch_restatement_headline_train_steps_f11(golden):
    """The oracle's training step at the HEADLINE configuration (M4B4, 16x3x48x48) against the
    reference's own train_step_larva (models/LarvaNet.py:98-114): 3 losses, all 82 gradients of
    step 1 (sampled values + |g| sums), weights after 3 AdamW steps."""
    g = golden("f11_m4b4_train_steps.npz")
    blocks = [4, 4, 4, 4]
    torch.set_num_threads(4)
    x, truth = _canonical_batch()
    sd = T.init_state_dict(blocks, seed=0)
    keys = [k[5:] for k in g.files if k.startswith("gidx.")]
    assert sorted(keys) == sorted(sd) and len(keys) == 82
    losses1, grads = T.train_steps(sd, x, truth, blocks, steps=1)
    for k in keys:
        gn = grads[k].numpy()
        tol = 2e-5 * float(g["gmax." + k])
        assert np.abs(gn.ravel()[g["gidx." + k]] - g["gval." + k]).max() <= tol, k
        assert abs(np.abs(gn.astype(np.float64)).sum() - float(g["gabs." + k])) <= 1e-4 * float(g["gabs." + k]), k
    np.testing.assert_allclose(losses1[0], g["losses"][0], rtol=1e-6)
    sd = T.init_state_dict(blocks, seed=0)
    losses, _ = T.train_steps(sd, x, truth, blocks, steps=3)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-6)
    flat = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    np.testing.assert_allclose(flat[::211], g["after3_sample"], rtol=0, atol=2e-6)


def test_torch_restatement_canonical_uint8_protocol_f12(golden):
    """Canonical forward -> validate.py uint8 protocol: image bytes and per-image PSNR against the
    synthetic truth as the reference produces them."""
    g = golden("f12_m4b4_uint8.npz")
    torch.set_num_threads(4)
    x, truth = _canonical_batch()
    sd = T.init_state_dict([4, 4, 4, 4], seed=0)
    with torch.no_grad():
        y = T.forward(sd, x, [4, 4, 4, 4]).numpy()
    y8 = np.stack([R.image_to_uint8(im) for im in y])
    t8 = np.stack([R.image_to_uint8(im) for im in truth.numpy()])
    assert np.array_equal(y8[0], g["u8_img0"]) and np.array_equal(y8[15], g["u8_img15"])
    psnr = np.array([R.image_psnr(y8[i], t8[i]) for i in range(16)])
    np.testing.assert_allclose(psnr, g["psnr_vs_truth"], rtol=0, atol=1e-6)


def _trajectory_batches(nb=4, seed0=1300):
    """The training batches of F13 (tests/golden/make_golden.py trajectory_batches): step s uses batch s mod nb."""
    pool = []
    for i in range(nb):
        g = torch.Generator().manual_seed(seed0 + i)
        pool.append((torch.rand(2, 3, 12, 12, generator=g) * 255, torch.rand(2, 3, 48, 48, generator=g) * 255))
    return pool


def _fake_val_pairs(seed=7):
    rng = np.random.RandomState(seed)
    pairs = []
    for (h, w) in ((10, 12), (9, 14)):
        lr = rng.randint(0, 256, size=(3, h, w)).astype(np.float32)
        hr = rng.randint(0, 256, size=(3, 4 * h + 1, 4 * w + 2)).astype(np.float32)
        pairs.append((lr, hr))
    return pairs


def test_torch_restatement_validation_trajectory_f13(golden):
    """F13: 72 steps of the reference's own train_step_larva through its validation branch (models/LarvaNet.py:116-137,
    141-161, 183-185): volumes, 25 validations, the plateau scheduler halving the learning rate at step 54, 24
    checkpoint names.  The oracle calls the same torch CPU operators, so everything is equal to the last bit but the
    PSNR the reference only PRINTS (8 decimals)."""
    g = golden("f13_val_trajectory.npz")
    torch.set_num_threads(4)
    sd = T.init_state_dict([2, 2], seed=0)
    rec = T.train_trajectory(sd, _trajectory_batches(), 72, [2, 2], _fake_val_pairs(), 400000000, 1.2e9, lr=2e-3)
    np.testing.assert_allclose(rec["losses"], g["losses"], rtol=1e-6)
    assert np.array_equal(rec["lrs"], g["lrs"]) and len(set(rec["lrs"])) == 2   # one halving
    assert rec["val_steps"] == list(g["val_steps"]) and rec["ckpt_names"] == list(g["ckpt_names"])
    np.testing.assert_allclose(rec["psnrs"], g["psnrs"], rtol=0, atol=2e-8)
    assert np.array_equal(rec["total_volume"], g["total_volume"]) and np.array_equal(rec["temp_volume"], g["temp_volume"])
    sch = rec["scheduler"]
    assert abs(sch.best - float(g["sched_best"])) < 1e-7
    assert (sch.num_bad_epochs, sch.cooldown_counter) == (int(g["sched_num_bad"]), int(g["sched_cooldown"]))
    flat = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    np.testing.assert_allclose(flat[::61], g["after_sample"], rtol=0, atol=1e-6)


def test_torch_restatement_v2_headline_steps_f14(golden):
    """F14: LarvaNetV2.train_step_larva (models/LarvaNetV2.py:101-148) at M4B4 on 16x3x48x48, three steps: losses,
    sampled gradients of all 88 tensors, weights after the third AdamW step (lr 1e-4, V2's default)."""
    g = golden("f14_v2_m4b4_train_steps.npz")
    torch.set_num_threads(8)
    blocks = [4, 4, 4, 4]
    sd = T.init_state_dict(blocks, v2=True, seed=3)
    flat0 = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    assert np.array_equal(flat0[::211], g["before_sample"])   # same init draw order as the reference's V2 module
    x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
    truth = torch.rand(16, 3, 192, 192, generator=torch.Generator().manual_seed(1)) * 255
    _, grads = T.train_steps(dict(sd), x, truth, blocks, steps=1, lr=1e-4, v2=True)
    assert len(grads) == 88
    for k, gr in grads.items():
        got = gr.numpy().ravel()[g["gidx." + k]]
        assert np.abs(got - g["gval." + k]).max() <= 1e-4 * max(float(g["gmax." + k]), 1e-30), k   # (summation order varies with the thread count)
    losses, _ = T.train_steps(sd, x, truth, blocks, steps=3, lr=1e-4, v2=True)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-6)
    flat = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    np.testing.assert_allclose(flat[::211], g["after3_sample"], rtol=0, atol=1e-6)


def test_edsr_restatement_train_steps_f15(golden):
    """BASELINE configs[0] (EDSR-baseline x4 on PyTorch-CPU via train.py; plumbing, no HIP kernels): oracle/edsr_torch.py
    against F15 = the reference's own EDSR.train_step (models/edsr.py:75-108) for 4 steps in train.py's call order,
    with the learning-rate decay firing after step 2: initial weights from the same seed bit for bit (including the two
    mean-shift layers, frozen RANDOM 1x1 convs: models/edsr.py:129-137 never installs their weights), losses,
    learning rates, weights after the 4th Adam step, the upscaled image, and the default network's key set / size."""
    from oracle import edsr_torch as E
    g = golden("f15_edsr_train_steps.npz")
    torch.set_num_threads(4)
    sd = _tsd(g)
    init = E.init_state_dict(16, 2, 4, seed=4)
    assert sorted(init) == sorted(sd) and all(torch.equal(init[k], sd[k]) for k in sd)
    step = E.make_trainer(sd, 2, lr_decay_steps=2)
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["truth"])
    losses, lrs = [], []
    for _ in range(4):
        losses.append(step(x, t))
        lrs.append(1e-4 * 0.5 ** ((step.state["global_step"] - 1) // 2))
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-6)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=0, atol=0)
    assert step.state["global_step"] == int(g["global_step"])
    for k in sd:
        np.testing.assert_allclose(step.params[k].detach().numpy(), g["after." + k], rtol=0, atol=1e-6)
        if k in E.FROZEN:
            assert np.array_equal(g["after." + k], g["sd." + k])      # the reference never trains them either
    with torch.no_grad():
        up = E.forward({k: v.detach() for k, v in step.params.items()}, x[:1], 2)[0].numpy()
    np.testing.assert_allclose(up[:, ::3, ::3], g["up_sample"], rtol=0, atol=1e-3)
    full = E.init_state_dict()
    assert sorted(full) == list(g["default_keys"]) and sum(v.numel() for v in full.values()) == int(g["default_params"])


def test_edsr_cpu_driver_loop_config0(tmp_path, capsys):
    """BASELINE configs[0] end to end on the CPU: the reference's train.py loop (train.py:83-105) over the EDSR
    restatement and the synthetic loader -- flag chaining, arguments.json, the log / save cadence, `model_%d.pth`
    names (models/edsr.py:60-62), the step-decayed learning rate, and a loss that goes down."""
    from oracle import train_edsr_cpu as D
    torch.manual_seed(0)
    np.random.seed(0)
    model, losses = D.main(["--train_path", str(tmp_path), "--max_steps=6", "--log_freq=2", "--save_freq=3", "--batch_size=2",
                            "--input_patch_size=12", "--synthetic_images=2", "--synthetic_lr_size=16", "--threads=4",
                            "--edsr_conv_features=16", "--edsr_res_blocks=2", "--edsr_learning_rate=1e-3",
                            "--edsr_learning_rate_decay_steps=4", "--bogus_flag=1"])
    out = capsys.readouterr().out
    assert model.global_step == 6 and len(losses) == 6
    assert "WARNING: found unhandled arguments: ['--bogus_flag=1']" in out
    assert out.count("step ") == 3 + 2 and "step 6, lr 0.000500" in out and "step 4, lr 0.001000" in out
    import json as _json
    import os as _os
    saved = sorted(n for n in _os.listdir(str(tmp_path)) if n.endswith(".pth"))
    assert saved == ["model_3.pth", "model_6.pth"]
    a = _json.load(open(_os.path.join(str(tmp_path), "arguments.json")))
    assert a["batch_size"] == 2 and a["edsr_res_blocks"] == 2 and a["max_steps"] == 6
    ck = torch.load(_os.path.join(str(tmp_path), "model_6.pth"))
    assert sorted(ck) == sorted(model.state_dict()) and losses[-1] < losses[0]
    up = model.upscale([np.zeros((3, 8, 9), np.float32)], 4)
    assert up.shape == (1, 3, 32, 36)


def _synthetic_task(steps=200, batch=4, patch=16, lr_size=40):
    """F16's / F17's data (tests/golden/make_golden.py synthetic_task): the repo's seeded, dataset-free loader."""
    from larvanet_amd.dataloaders import synthetic_loader as S
    tr = S.create_loader()
    tr.parse_args(["--synthetic_images=6", "--synthetic_lr_size=%d" % lr_size, "--data_seed=3"])
    tr.prepare([4])
    batches = []
    for _ in range(steps):
        x, t = tr.get_patch_batch(batch, 4, patch)
        batches.append((torch.from_numpy(np.stack(x)), torch.from_numpy(np.stack(t))))
    val = S.create_loader()
    val.parse_args(["--synthetic_images=3", "--synthetic_lr_size=32", "--data_seed=9"])
    val.prepare([4])
    return batches, val


def test_torch_restatement_realistic_training_f16(golden):
    """F16: 200 steps of the reference's train_step_larva at its default learning rate on a learnable task (smooth
    synthetic images, LR = box-filtered HR), 9 validations: the oracle's PSNR trajectory within 0.005 dB of the
    reference's (the reference re-run one ulp away: 0.0018 dB), losses within 1e-3."""
    g = golden("f16_realistic_training.npz")
    torch.set_num_threads(4)
    batches, val = _synthetic_task()
    pairs = [val.get_image_pair(i, 4)[:2] for i in range(val.get_num_images())]
    sd = T.init_state_dict([2, 2], seed=0)
    rec = T.train_trajectory(sd, batches, 200, [2, 2], pairs, 1, 25, lr=4e-4)
    assert len(rec["psnrs"]) == len(g["psnrs"]) == 9 and rec["lrs"] == list(g["lrs"])
    np.testing.assert_allclose(rec["psnrs"], g["psnrs"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(rec["losses"], g["losses"], rtol=1e-3)
    assert g["psnrs"][-1] > g["psnrs"][0] + 0.03    # (the task is learnable: the reference's PSNR goes up)


def test_torch_restatement_headline_training_f17_first_steps(golden):
    """F17 (the reference's 60-step run at the headline configuration): the oracle on the first 4 steps and the first
    validation -- the whole run is the GPU test's job (tests/test_headline_parity.py); on the CPU it costs 25 s."""
    g = golden("f17_headline_training.npz")
    torch.set_num_threads(8)
    batches, val = _synthetic_task(steps=4, batch=16, patch=48, lr_size=96)
    pairs = [val.get_image_pair(i, 4)[:2] for i in range(val.get_num_images())]
    sd = T.init_state_dict([4, 4, 4, 4], seed=0)
    rec = T.train_trajectory(sd, batches, 4, [4, 4, 4, 4], pairs, 1, 20, lr=4e-4)
    np.testing.assert_allclose(rec["losses"], g["losses"][:4], rtol=1e-4)
    assert abs(rec["psnrs"][0] - g["psnrs"][0]) < 2e-3
