"""Parity at the HEADLINE configuration -- the workload bench.py reports (BASELINE config 2/3:
LarvaNet x4, --num_modules=4 --num_blocks=4,4,4,4, batch 16 x 3 x 48 x 48 -> 16 x 3 x 192 x 192)
and the full-image inference of config 5 (3 x 339 x 510, V1 and V2).  At this size the training
step issues the kernels in the shapes the benchmark times: wgrad3x3_pipe_kernel<48,48> as
32 layers x 8 workgroups + 8 x 32, ExitsFn with 4 batched jobs of 256 workgroups, the K = 96
JointBwd dgrad and the deferred weight-gradient queue.

Checked against (1) fixtures generated from the imported reference (tests/golden/make_golden.py r2:
F11 = its own train_step_larva for 3 steps, F12 = its forward after the validate.py uint8
protocol) and (2) oracle/larva_torch.py (pinned against the same fixtures on the CPU by
tests/test_oracle_golden.py) for every gradient element.

Tolerances: losses 2e-5 relative; gradients 2e-4 of each tensor's largest element; fp32 forward
2e-3 absolute on the 0-255 scale; PSNR after the uint8 protocol 1e-3 dB (north_star)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BLOCKS = [4, 4, 4, 4]
FLAGS = ["--num_modules=4", "--num_blocks=4,4,4,4"]


def _model(name, argv, training=False, seed=0):
    import importlib
    m = importlib.import_module("larvanet_amd.models." + name).create_model()
    m.parse_args(argv)
    torch.manual_seed(seed)
    m.prepare(is_training=training, scales=[4])
    return m


class FakeValLoader:
    """Same two pairs as tests/golden/make_golden.py: validate_for_train at global_step 1."""

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


def _canonical_batch():
    x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
    truth = torch.rand(16, 3, 192, 192, generator=torch.Generator().manual_seed(1)) * 255
    return x, truth


@pytest.fixture(scope="module")
def oracle_step():
    """One training step of the oracle at the headline size: loss + all 82 gradients (CPU, < 1 s)."""
    from oracle import larva_torch as T
    x, truth = _canonical_batch()
    sd = T.init_state_dict(BLOCKS, seed=0)
    losses, grads = T.train_steps(dict(sd), x, truth, BLOCKS, steps=1)
    return losses[0], grads


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_headline_train_step_matches_reference_fixture_f11(hip_device, golden, oracle_step, use_graph):
    """(a) + (b): the plugin's train_step_larva at M4B4 / 16x3x48x48 against the reference's own
    three steps (F11) and, element for element, against the oracle's gradients."""
    g = golden("f11_m4b4_train_steps.npz")
    ref_loss, ref_grads = oracle_step
    m = _model("LarvaNet", FLAGS, training=True)
    m.use_hip_graph = use_graph
    m.volume_per_step = 48 * 48 * 16 * 3
    x, truth = _canonical_batch()
    x, truth = x.to(hip_device), truth.to(hip_device)
    args = types.SimpleNamespace(train_path="/tmp")
    val = FakeValLoader(7)
    losses = []
    for step in range(3):
        losses.append(m.train_step_larva(args, val, x, truth, None))
        if step == 0:
            assert abs(losses[0] - ref_loss) <= 2e-5 * abs(ref_loss)
            names = [k for k, _ in m.model.named_parameters()]
            assert len(names) == 82
            for k, p in m.model.named_parameters():
                got = p.grad.detach().cpu().numpy()
                # the reference's own values (sampled) ...
                tol = 2e-4 * max(float(g["gmax." + k]), 1e-30)
                d = float(np.abs(got.ravel()[g["gidx." + k]] - g["gval." + k]).max())
                assert d <= tol, ("fixture", k, d, tol)
                gabs = float(np.abs(got.astype(np.float64)).sum())
                assert abs(gabs - float(g["gabs." + k])) <= 2e-4 * float(g["gabs." + k]), ("fixture |g|", k)
                # ... and every element against the oracle
                ref = ref_grads[k].numpy()
                d = float(np.abs(got - ref).max())
                assert d <= 2e-4 * max(float(np.abs(ref).max()), 1e-30), ("oracle", k, d)
    assert m.use_hip_graph == use_graph  # a capture failure would have switched it off silently
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    after = {k: v.cpu().numpy() for k, v in m.model.state_dict().items()}
    flat = np.concatenate([after[k].ravel() for k in sorted(after)])
    np.testing.assert_allclose(flat[::211], g["after3_sample"], rtol=0, atol=2e-5)
    assert m.global_step == int(g["global_step"]) and m.temp_volume == int(g["temp_volume"])
    np.testing.assert_allclose(m.get_lr(), g["lrs"][-1])


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_headline_train_steps_v2_match_reference_fixture_f14(hip_device, golden, use_graph):
    """LarvaNetV2 (models/LarvaNetV2.py:101-148, 314-365) at the headline size -- M4B4, 16 x 3 x 48 x 48, so the merge
    conv reads four 7 MB feature tensors as K = 192 and the tail is a fifth exit: three train_step_larva steps
    against F14 = the reference's own three steps from the same weights (seed 3; lr 1e-4, V2's default): every
    loss, sampled values and |g| sums of all 88 first-step gradients, the weights after the third AdamW step; and
    every gradient element against oracle/larva_torch.py (pinned to F14 on the CPU)."""
    from oracle import larva_torch as T
    g = golden("f14_v2_m4b4_train_steps.npz")
    m = _model("LarvaNetV2", FLAGS, training=True, seed=3)
    m.use_hip_graph = use_graph
    m.volume_per_step = 48 * 48 * 16 * 3
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    flat0 = np.concatenate([sd[k].numpy().ravel() for k in sorted(sd)])
    assert np.array_equal(flat0[::211], g["before_sample"])    # the reference's initial weights, bit for bit
    x, truth = _canonical_batch()
    _, ref_grads = T.train_steps(dict(sd), x, truth, BLOCKS, steps=1, lr=m.get_lr(), v2=True)
    args = types.SimpleNamespace(train_path="/tmp")
    xd, td = x.to(hip_device), truth.to(hip_device)
    losses, worst = [], {"fixture": 0.0, "oracle": 0.0, "gabs": 0.0, "fixture64": 0.0}
    for step in range(3):
        losses.append(m.train_step_larva(args, FakeValLoader(7), xd, td, None))
        if step == 0:
            assert len(list(m.model.named_parameters())) == 88
            for k, p in m.model.named_parameters():
                got = p.grad.detach().cpu().numpy()
                gmax = max(float(g["gmax." + k]), 1e-30)
                worst["fixture"] = max(worst["fixture"], float(np.abs(got.ravel()[g["gidx." + k]] - g["gval." + k]).max()) / gmax)
                worst["fixture64"] = max(worst["fixture64"], float(np.abs(got.ravel()[g["gidx." + k]] - g["gval64." + k]).max()) / gmax)
                worst["oracle"] = max(worst["oracle"], float(np.abs(got - ref_grads[k].numpy()).max()) / gmax)
                gabs = float(np.abs(got.astype(np.float64)).sum())
                worst["gabs"] = max(worst["gabs"], abs(gabs - float(g["gabs." + k])) / float(g["gabs." + k]))
    own = float(g["ref32_vs_ref64_worst"])
    print("F14 %s: gradient deviations (of each tensor's max): %s; the reference's own fp32 run against its fp64 run: %.2e"
          % ("hipgraph" if use_graph else "eager", worst, own))
    # Bars.  DESIGN section 6's gradient bar is 2e-4 of each tensor's largest element, and V1 (F11) meets it.  Here it
    # cannot be met BY THE REFERENCE ITSELF: the L1 gradient is sign(out - truth), of the 8.8 M (output, truth) pairs of
    # the five exits a handful lie closer together than the forward's rounding error, and a flipped sign moves single
    # weight-gradient elements by a few 1e-4 of the tensor's maximum -- F14 holds the reference's first step in float64
    # as well, and its own fp32 gradients sit 4.9e-4 from those.  So: every element within 5e-4 of the fp32 reference
    # / oracle, no sampled element further from the EXACT (fp64) gradient than the reference's own fp32 run is
    # (measured 2.5e-4 / 3.1e-4 / see the printed line), each tensor's |g| sum within 2e-4 (measured 2.4e-5).
    assert worst["fixture"] <= 5e-4 and worst["oracle"] <= 5e-4 and worst["gabs"] <= 2e-4, worst
    assert worst["fixture64"] <= max(own, 2e-4), (worst, own)
    assert m.use_hip_graph == use_graph
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    np.testing.assert_allclose(m.get_lr(), g["lrs"][-1])
    assert m.global_step == int(g["global_step"]) and m.temp_volume == int(g["temp_volume"])
    # AdamW's first steps move a weight by lr * g / (|g| + eps): where |g| is of the order of eps = 1e-8 the move depends
    # on g's last bits, so single elements may sit up to steps * lr apart; everything else within 2e-5
    after = {k: v.cpu().numpy() for k, v in m.model.state_dict().items()}
    flat = np.concatenate([after[k].ravel() for k in sorted(after)])
    d = np.abs(flat[::211] - g["after3_sample"])
    assert float((d > 2e-5).mean()) < 1e-3 and float(d.max()) <= 3.1 * m.get_lr(), (float(d.max()), float((d > 2e-5).mean()))


def _trajectory_batches(nb=4, seed0=1300):
    """The training batches of F13 (tests/golden/make_golden.py trajectory_batches): step s uses batch s mod nb."""
    pool = []
    for i in range(nb):
        gen = torch.Generator().manual_seed(seed0 + i)
        pool.append((torch.rand(2, 3, 12, 12, generator=gen) * 255, torch.rand(2, 3, 48, 48, generator=gen) * 255))
    return pool


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_validation_branch_trajectory_matches_reference_fixture_f13(hip_device, golden, tmp_path, capsys, use_graph):
    """F13: the plugin's train_step_larva driven for 72 steps through the reference's validation branch
    (models/LarvaNet.py:116-137: temp_volume >= val_volume -> total_volume bookkeeping -> validate_for_train ->
    scheduler.step(avg_psnr) at :161 -> save() name at :183-185) against what the reference itself did from the same
    weights on the same four cycling batches: 25 validations, ReduceLROnPlateau (patience 3, cooldown 6) halving the
    learning rate after the validation of step 54, 24 checkpoints.

    What must be EXACT: the learning-rate sequence (i.e. every plateau decision), the validation steps, both volume
    counters after every step, the checkpoint file names, the scheduler's counters at the end.
    What drifts: fp32 summation order differs between the MFMA kernels and ATen's CPU convolution (1e-6 relative per
    layer), and 72 AdamW steps at lr 2e-3 amplify rounding-level differences: the REFERENCE re-run from initial weights
    one ulp away (F13's ulp_tube_*) stays within 0.012 dB / 1.2e-3 relative loss of itself up to step 51 and then
    jumps to 0.12 dB / 1.5e-2 at the loss spike of steps 54-59.  Measured on MI355X (printed below): the same picture --
    max 1.5e-2 on the loss at step 59, 1.0e-3 at step 72, 0.12 dB at the validation of step 54.  Bars: see (a), (b)."""
    g = golden("f13_val_trajectory.npz")
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2", "--lr=2e-3", "--val_volume=1.2e9"], training=True)
    m.use_hip_graph = use_graph
    m.volume_per_step = 400000000
    args = types.SimpleNamespace(train_path=str(tmp_path))
    val = FakeValLoader(7)
    pool = [(a.to(hip_device), b.to(hip_device)) for a, b in _trajectory_batches()]
    losses, lrs, tot, tmp, val_steps, psnrs = [], [], [], [], [], []
    capsys.readouterr()
    for step in range(72):
        x, t = pool[step % len(pool)]
        losses.append(m.train_step_larva(args, val, x, t, None))
        lrs.append(m.get_lr())
        tot.append(m.total_volume)
        tmp.append(m.temp_volume)
        for line in capsys.readouterr().out.splitlines():
            if "psnr=" in line:
                val_steps.append(m.global_step)
                psnrs.append(float(line.split("psnr=")[1].split(",")[0]))
    assert m.use_hip_graph == use_graph
    loss_dev = np.abs(np.array(losses) / g["losses"] - 1)
    psnr_dev = np.abs(np.array(psnrs) - g["psnrs"]) if len(psnrs) == len(g["psnrs"]) else None
    with capsys.disabled():
        print("\nF13 %s: max relative loss deviation %.2e (step %d; at step 72 %.2e), max PSNR deviation %s dB" % (
            "hipgraph" if use_graph else "eager", loss_dev.max(), int(loss_dev.argmax()) + 1, loss_dev[-1],
            None if psnr_dev is None else "%.2e" % psnr_dev.max()))
    assert val_steps == list(g["val_steps"])
    assert np.array_equal(np.array(lrs), g["lrs"]), [i + 1 for i in range(72) if lrs[i] != g["lrs"][i]]
    assert np.array_equal(np.array(tot, np.float64), g["total_volume"]) and np.array_equal(np.array(tmp, np.float64), g["temp_volume"])
    import os
    names = sorted((n for n in os.listdir(str(tmp_path)) if n.startswith("model_step")), key=lambda n: int(n.split("_")[1][4:]))
    assert names == list(g["ckpt_names"])
    sch = m.scheduler
    assert (sch.num_bad_epochs, sch.cooldown_counter) == (int(g["sched_num_bad"]), int(g["sched_cooldown"]))
    assert abs(sch.best - float(g["sched_best"])) < 0.01
    # (a) before the trajectory turns chaotic (the reference's own one-ulp tube is < 0.012 dB wide up to step 51):
    #     north_star's 0.02 dB and 2e-3 relative on the loss
    early = np.array(val_steps) <= 48
    assert loss_dev[:48].max() < 2e-3 and psnr_dev[early].max() < 0.02, (loss_dev[:48].max(), psnr_dev[early].max())
    # (b) all 72 steps: inside (3 x) the tube the REFERENCE sweeps out when its initial weights move by one fp32 ulp
    #     (F13's ulp_tube_*: four perturbed runs of the reference, running maximum of the deviation; it jumps to 0.12 dB /
    #     1.5e-2 at the loss spike of steps 54-59 -- every one of those runs takes the same plateau decisions)
    assert bool(g["ulp_tube_same_lrs"])
    assert (loss_dev <= 3 * g["ulp_tube_loss"] + 1e-5).all(), np.argwhere(loss_dev > 3 * g["ulp_tube_loss"] + 1e-5).ravel()
    assert (psnr_dev <= 3 * g["ulp_tube_psnr"] + 2e-3).all(), np.argwhere(psnr_dev > 3 * g["ulp_tube_psnr"] + 2e-3).ravel()
    # the last checkpoint is the reference-format bare state_dict of the final weights
    last = torch.load(os.path.join(str(tmp_path), names[-1]), map_location="cpu")
    flat = np.concatenate([last[k].numpy().ravel() for k in sorted(last)])
    d = np.abs(flat[::61] - g["after_sample"])
    with capsys.disabled():
        print("F13 weights after 72 steps: max |dw| %.2e, mean %.2e (weights are O(0.01-0.1))" % (d.max(), d.mean()))
    assert d.mean() < 3 * float(g["ulp_tube_weights_mean"])


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "hipgraph"])
def test_realistic_training_psnr_within_north_star_of_reference_f16(hip_device, golden, tmp_path, capsys, use_graph):
    """north_star: "PSNR within 0.02 dB of reference" (training quality).  DIV2K cannot be had here; the closest
    obtainable stand-in: F16 = the REFERENCE's own train_step_larva for 200 steps at its default learning rate (4e-4) on a
    learnable task -- the repo's seeded synthetic loader (smooth images, LR = box-filtered HR), 200 different batches of
    4 x 3 x 16 x 16, validation on 3 synthetic images at step 1 and every 25 steps (9 validations, 8 checkpoints).  The
    plugin, from the same initial weights on the same batches, must stay within 0.02 dB of the reference at EVERY
    validation and within 2e-3 relative on every loss; measured (printed): a few 1e-3 dB -- the reference re-run one
    ulp away from itself moves 1.8e-3 dB."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_golden import _synthetic_task
    g = golden("f16_realistic_training.npz")
    m = _model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2", "--val_volume=25"], training=True)
    m.use_hip_graph = use_graph
    m.volume_per_step = 1
    batches, val = _synthetic_task()
    args = types.SimpleNamespace(train_path=str(tmp_path))
    losses, lrs, psnrs = [], [], []
    capsys.readouterr()
    for x, t in batches:
        losses.append(m.train_step_larva(args, val, x.to(hip_device), t.to(hip_device), None))
        lrs.append(m.get_lr())
        psnrs += [float(line.split("psnr=")[1].split(",")[0]) for line in capsys.readouterr().out.splitlines() if "psnr=" in line]
    assert m.use_hip_graph == use_graph
    psnr_dev = np.abs(np.array(psnrs) - g["psnrs"])
    loss_dev = np.abs(np.array(losses) / g["losses"] - 1)
    with capsys.disabled():
        print("\nF16 %s: PSNR %.4f -> %.4f dB (reference %.4f -> %.4f), max deviation %.2e dB (reference one ulp away: %.2e), "
              "max relative loss deviation %.2e" % ("hipgraph" if use_graph else "eager", psnrs[0], psnrs[-1], g["psnrs"][0],
                                                    g["psnrs"][-1], psnr_dev.max(), g["ulp_tube_psnr"].max(), loss_dev.max()))
    assert lrs == list(g["lrs"]) and len(psnrs) == 9
    assert psnr_dev.max() < 0.02, psnr_dev
    assert loss_dev.max() < 2e-3, loss_dev.max()
    assert len([n for n in os.listdir(str(tmp_path)) if n.startswith("model_step")]) == 8


def test_headline_size_training_psnr_within_north_star_of_reference_f17(hip_device, golden, tmp_path, capsys):
    """F16's question at the HEADLINE configuration -- M4B4, 16 x 3 x 48 x 48 per step, i.e. the launch geometry bench.py
    times (two half-batch strip chains of 256 workgroups, batched exits scored in their conv launch, the flat 40-layer
    weight-gradient grid with the head as its tail, one-launch AdamW, the captured graph): F17 = the reference's own 60
    steps on the learnable synthetic task, validation at step 1 and every 20 steps.  Bars: every validation within 0.02 dB
    (north_star), every loss within 2e-3 relative; the reference re-run one ulp away moves 4.5e-3 dB / 1.3e-4."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_golden import _synthetic_task
    g = golden("f17_headline_training.npz")
    m = _model("LarvaNet", FLAGS + ["--val_volume=20"], training=True)
    m.volume_per_step = 1
    batches, val = _synthetic_task(steps=60, batch=16, patch=48, lr_size=96)
    args = types.SimpleNamespace(train_path=str(tmp_path))
    losses, psnrs = [], []
    capsys.readouterr()
    for x, t in batches:
        losses.append(m.train_step_larva(args, val, x.to(hip_device), t.to(hip_device), None))
        psnrs += [float(line.split("psnr=")[1].split(",")[0]) for line in capsys.readouterr().out.splitlines() if "psnr=" in line]
    assert m.use_hip_graph and m.hip_graph_fell_back is None
    psnr_dev = np.abs(np.array(psnrs) - g["psnrs"])
    loss_dev = np.abs(np.array(losses) / g["losses"] - 1)
    with capsys.disabled():
        print("\nF17: PSNR %.4f -> %.4f dB (reference %.4f -> %.4f), max deviation %.2e dB (reference one ulp away: %.2e), "
              "max relative loss deviation %.2e (reference: %.2e)" % (psnrs[0], psnrs[-1], g["psnrs"][0], g["psnrs"][-1],
                                                                      psnr_dev.max(), g["ulp_tube_psnr"].max(), loss_dev.max(),
                                                                      g["ulp_tube_loss"].max()))
    assert len(psnrs) == 4 and psnr_dev.max() < 0.02, psnr_dev
    assert loss_dev.max() < 2e-3, loss_dev.max()


def test_headline_wgrad_launch_shape_32_layers_by_8_splits(hip_device):
    """The weight-gradient launch exactly as the step issues it (32 layers x 8 workgroups at
    16x48x48x48, pipelined kernel + fixed-order reduction) against torch's CPU conv2d_weight."""
    from larvanet_amd import kernels as K
    gen = torch.Generator().manual_seed(31)
    jobs, refs = [], []
    for i in range(32):
        dy = torch.randn(16, 48, 48, 48, generator=gen) * 1e-3
        x = torch.randn(16, 48, 48, 48, generator=gen) * 20
        jobs.append({"dy": dy.to(hip_device), "x": x.to(hip_device),
                     "dw": torch.full((48, 48, 3, 3), float("nan"), device=hip_device),
                     "db": torch.full((48,), float("nan"), device=hip_device)})
        if i in (0, 13, 31):   # float64 references for three of the layers (the others: fp32 CPU)
            dw = torch.nn.grad.conv2d_weight(x.double(), (48, 48, 3, 3), dy.double(), padding=1)
            refs.append((i, dw.float().numpy(), dy.double().sum((0, 2, 3)).float().numpy(), 3e-5))
        else:
            dw = torch.nn.grad.conv2d_weight(x, (48, 48, 3, 3), dy, padding=1)
            refs.append((i, dw.numpy(), dy.sum((0, 2, 3)).numpy(), 2e-4))
    K.conv3x3_wgrad(jobs, 48, 48, 8)
    torch.cuda.synchronize()
    for i, dw_ref, db_ref, rel in refs:
        dw, db = jobs[i]["dw"].cpu().numpy(), jobs[i]["db"].cpu().numpy()
        assert np.isfinite(dw).all() and np.isfinite(db).all()
        assert np.abs(dw - dw_ref).max() <= rel * np.abs(dw_ref).max(), i
        assert np.abs(db - db_ref).max() <= rel * max(np.abs(db_ref).max(), 1e-30) + 1e-7, i


def test_canonical_forward_uint8_protocol_f6_f12(hip_device, golden):
    """(d): canonical-size inference forward: sampled fp32 values of the reference's output within
    2e-3 (F6), and after the validate.py protocol (round half to even, clip, uint8) the image the
    reference produces (F12): per-image PSNR against the synthetic truth within 1e-3 dB, uint8
    pixels differing from the reference's only where fp32 summation order crosses a .5 boundary."""
    from larvanet_amd import metrics
    g6, g12 = golden("f6_m4b4_canonical.npz"), golden("f12_m4b4_uint8.npz")
    m = _model("LarvaNet", FLAGS)
    x, truth = _canonical_batch()
    with torch.no_grad():
        y = m.model(x.to(hip_device)).cpu().numpy()
    d = np.abs(y.ravel()[g6["sample_idx"]] - g6["sample_val"])
    assert float(d.max()) < 2e-3, float(d.max())
    np.testing.assert_allclose(y[0][:, ::3, ::3], g12["out_img0_f32"], rtol=0, atol=2e-3)
    y8 = np.stack([metrics.image_to_uint8(im) for im in y])
    t8 = np.stack([metrics.image_to_uint8(im) for im in truth.numpy()])
    psnr = np.array([float(metrics.image_psnr(y8[i], t8[i])) for i in range(16)])
    assert np.abs(psnr - g12["psnr_vs_truth"]).max() < 1e-3, np.abs(psnr - g12["psnr_vs_truth"]).max()
    for idx, key in ((0, "u8_img0"), (15, "u8_img15")):
        diff = np.abs(y8[idx].astype(np.int16) - g12[key].astype(np.int16))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, (idx, int(diff.max()), float((diff != 0).mean()))
        # our image scored against the reference's image: > 80 dB means "the same picture"
        assert float(metrics.image_psnr(y8[idx], g12[key])) > 80.0
    # the device-side PSNR kernel (validate_for_train's path) gives the host protocol's number
    from larvanet_amd import kernels as K
    out0 = torch.from_numpy(y[3]).to(hip_device)
    # (exact integer sum on the device, float32 mean on the host: equal to ~1e-7 dB)
    assert abs(K.psnr_u8(out0, torch.from_numpy(t8[3]).to(hip_device)) - psnr[3]) < 1e-5


@pytest.mark.parametrize("name", ["LarvaNet", "LarvaNetV2"])
@pytest.mark.parametrize("staging", ["pitched", "scalar"])
def test_full_image_upscale_339x510_against_oracle(hip_device, name, staging):
    """(c): BASELINE config 5 at N = 1: whole-network `upscale` of a DIV2K-val-sized LR image
    (3 x 339 x 510 -> 3 x 1356 x 2040), V1 and V2, on both staging paths of the conv kernel (510 is
    not a multiple of 4: `pitched` pads the rows to 512 and keeps the 16-byte LDS-DMA path, `scalar`
    runs the register-staged path on the unpadded tensors), against oracle/larva_torch.py."""
    from oracle import larva_torch as T
    from larvanet_amd import metrics
    v2 = name == "LarvaNetV2"
    m = _model(name, FLAGS, seed=0)
    m.model.pad_odd_widths = staging == "pitched"
    rng = np.random.RandomState(510)
    img = rng.randint(0, 256, size=(3, 339, 510)).astype(np.float32)
    got = m.upscale([img], 4)[0]
    assert got.shape == (3, 1356, 2040)
    sd = {k: v.detach().cpu() for k, v in m.model.state_dict().items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        xt = torch.from_numpy(img)[None]
        ref = (T.forward_v2(sd, xt, BLOCKS) if v2 else T.forward(sd, xt, BLOCKS))[0].numpy()
    assert float(np.abs(got - ref).max()) <= 2e-3, float(np.abs(got - ref).max())
    g8, r8 = metrics.image_to_uint8(got), metrics.image_to_uint8(ref)
    diff = np.abs(g8.astype(np.int16) - r8.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    hr = rng.randint(0, 256, size=(3, 1356, 2040)).astype(np.uint8)
    assert abs(float(metrics.image_psnr(g8, hr)) - float(metrics.image_psnr(r8, hr))) < 1e-3


def test_headline_v2_train_step_against_oracle(hip_device):
    """LarvaNetV2 (tail exit over the un-materialised concatenation of 4 features, K = 192 merge
    conv, (M+1)-way loss) at the headline batch: loss and all 88 gradients against the oracle."""
    from oracle import larva_torch as T
    m = _model("LarvaNetV2", FLAGS, training=True, seed=0)
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    x, truth = _canonical_batch()
    loss, _ = m._forward_backward(x.to(hip_device), truth.to(hip_device))
    m._finish_backward()
    torch.cuda.synchronize()
    sd_req = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(8)
    ref_loss = T.multi_exit_loss(sd_req, x, truth, BLOCKS, v2=True)
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 2e-5 * abs(float(ref_loss.detach()))
    n = 0
    for k, p in m.model.named_parameters():
        ga, gb = p.grad.cpu().numpy(), sd_req[k].grad.numpy()
        assert np.abs(ga - gb).max() <= 2e-4 * max(np.abs(gb).max(), 1e-30), k
        n += 1
    assert n == 88


@pytest.mark.parametrize("nf", [32, 64])
def test_train_step_at_the_timed_size_with_num_filters_32_and_64(hip_device, nf):
    """bench.py's `other_widths` times `train_step_larva` of the M4B4 network built with --num_filters=32 / 64 on the
    headline batch (BASELINE configs 2 / 5 read literally; a build-side extension).  Until round 4 those
    networks were value-checked only at M2 on 4 x 3 x 12 x 16.  Here: the captured (hipGraph) step at M4B4 on
    16 x 3 x 48 x 48 -- the flat (nf, nf) weight-gradient grid over 40 layers, the (48, nf) leg gradients, the
    nf-channel strip chains of the DualChain -- loss and EVERY gradient element against oracle/larva_torch.py built at
    the same width from the same seed (weights bit-identical, asserted).  Gradient bar 5e-4 of each tensor's maximum, as
    for F14: the L1 gradient is sign(out - truth), a handful of the 8.8 M (exit, pixel) pairs lie within the forward's
    rounding error of each other, and a flipped sign moves single weight-gradient elements by more than 2e-4 (the
    reference's own fp32 run sits 4.9e-4 from its float64 run at this size); |g| sums within 2e-4."""
    from oracle import larva_torch as T
    m = _model("LarvaNet", FLAGS + ["--num_filters=%d" % nf], training=True, seed=0)
    assert m.use_hip_graph and m.dual_chain
    sd = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
    ref_sd = T.init_state_dict(BLOCKS, seed=0, nf=nf)
    assert sorted(sd) == sorted(ref_sd) and all(torch.equal(sd[k], ref_sd[k]) for k in sd)
    x, truth = _canonical_batch()
    torch.set_num_threads(8)
    ref_losses, ref_grads = T.train_steps(dict(sd), x, truth, BLOCKS, steps=1, lr=m.get_lr())
    m.volume_per_step = 48 * 48 * 16 * 3
    args = types.SimpleNamespace(train_path="/tmp")
    loss = m.train_step_larva(args, FakeValLoader(7), x.to(hip_device), truth.to(hip_device))
    assert m.use_hip_graph and m.hip_graph_fell_back is None   # the captured step ran, not an eager fall-back
    assert abs(loss - ref_losses[0]) <= 2e-5 * abs(ref_losses[0])
    n = 0
    for k, p in m.model.named_parameters():
        got, ref = p.grad.detach().cpu().numpy(), ref_grads[k].numpy()
        # (absolute floor: the exits' bias gradients are sums of +-g, g = 1 / (4 * 16 * 3 * 192 * 192) = 1.4e-7, that nearly
        # cancel -- ONE flipped sign moves such an element by 2 g = 2.8e-7, 1.7e-3 of a tensor maximum of 1.6e-4)
        assert np.abs(got - ref).max() <= max(5e-4 * np.abs(ref).max(), 1e-6), k
        ga, gb = float(np.abs(got.astype(np.float64)).sum()), float(np.abs(ref.astype(np.float64)).sum())
        assert abs(ga - gb) <= 2e-4 * gb + 1e-6, ("|g| sum", k)
        n += 1
    assert n == 82


def test_full_image_upscale_339x510_v2_with_64_filters_against_oracle(hip_device):
    """BASELINE configs[4] read literally -- "LarvaNetV2 x4, 64ch body, full-image inference" -- is what bench.py times
    as infer_full_image.LarvaNetV2_64ch: `upscale` of a 3 x 339 x 510 image through the --num_filters=64 V2 network
    (wide 64-channel tiles on the pitched rows, the K = 256 merge conv over four feature tensors, the 64 -> 48 tail),
    against oracle/larva_torch.py at that width: fp32 output, uint8 protocol, PSNR."""
    from oracle import larva_torch as T
    from larvanet_amd import metrics
    m = _model("LarvaNetV2", FLAGS + ["--num_filters=64"], seed=0)
    rng = np.random.RandomState(511)
    img = rng.randint(0, 256, size=(3, 339, 510)).astype(np.float32)
    got = m.upscale([img], 4)[0]
    assert got.shape == (3, 1356, 2040)
    sd = {k: v.detach().cpu() for k, v in m.model.state_dict().items()}
    ref_sd = T.init_state_dict(BLOCKS, v2=True, seed=0, nf=64)
    assert all(torch.equal(sd[k], ref_sd[k]) for k in sd)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = T.forward_v2(sd, torch.from_numpy(img)[None], BLOCKS)[0].numpy()
    assert float(np.abs(got - ref).max()) <= 2e-3, float(np.abs(got - ref).max())
    g8, r8 = metrics.image_to_uint8(got), metrics.image_to_uint8(ref)
    diff = np.abs(g8.astype(np.int16) - r8.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    hr = rng.randint(0, 256, size=(3, 1356, 2040)).astype(np.uint8)
    assert abs(float(metrics.image_psnr(g8, hr)) - float(metrics.image_psnr(r8, hr))) < 1e-3
