"""Generates tests/golden/*.npz by IMPORTING the reference (read-only at /root/reference) in the
build container.  The reference cannot travel to the GPU box; these small input/output vectors
can.  Run:  python tests/golden/make_golden.py

Accommodations needed to import the reference here (both ordinary Python errors, see SURVEY.md
section 8c): an empty stand-in module for `cv2` (imported by validate.py:14, never called on this
path), and prepare(is_training=False) + hand-attached loss/optimizer/scheduler because
ReduceLROnPlateau(verbose=...) no longer exists in torch 2.10 (models/LarvaNet.py:90-92).
"""
import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# LARVA_GOLDEN_OUT=<dir>: write the fixtures there instead of beside this script (to check that the committed ones
# regenerate bit for bit:  LARVA_GOLDEN_OUT=/tmp/g python tests/golden/make_golden.py [r2|r3]  then compare the arrays)
OUT = os.environ.get("LARVA_GOLDEN_OUT", HERE)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def make_ref_model(name, argv, seed):
    mod = importlib.import_module("models." + name)
    model = mod.create_model()
    model.parse_args(argv)
    torch.manual_seed(seed)
    model.prepare(is_training=False, scales=[4])
    return model


def attach_training(model):
    import torch.nn as nn
    import torch.optim as optim
    model.loss_fn = nn.L1Loss()
    model.optim = optim.AdamW(filter(lambda p: p.requires_grad, model.model.parameters()), lr=model.args.lr)
    model.scheduler = optim.lr_scheduler.ReduceLROnPlateau(
        model.optim, mode="max", factor=model.args.lr_decay, patience=model.args.patience,
        cooldown=getattr(model.args, "cooldown", 0), threshold=model.args.threshold, threshold_mode="abs",
        min_lr=model.args.min_lr)


class FakeValLoader:
    """Two tiny synthetic validation pairs (uint8-valued), enough for validate_for_train."""

    def __init__(self, seed):
        rng = np.random.RandomState(seed)
        self.pairs = []
        for (h, w) in ((10, 12), (9, 14)):
            lr = rng.randint(0, 256, size=(3, h, w)).astype(np.float32)
            hr = rng.randint(0, 256, size=(3, 4 * h + 1, 4 * w + 2)).astype(np.float32)  # larger: exercises the crop
            self.pairs.append((lr, hr))

    def get_num_images(self):
        return len(self.pairs)

    def get_image_pair(self, image_index, scale):
        lr, hr = self.pairs[image_index]
        return lr, hr, "img%d" % image_index


def main_r2(only):
    """Round-2 fixtures (headline configuration): F11 = the reference's own train_step_larva at
    M4B4 on 16x3x48x48 (models/LarvaNet.py:98-114), F12 = the canonical forward after the
    validate.py uint8 protocol (validate.py:17-27).  `python make_golden.py r2` writes only these."""
    validate = importlib.import_module("validate")
    argv = ["--num_modules=4", "--num_blocks=4,4,4,4"]
    x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
    truth = torch.rand(16, 3, 192, 192, generator=torch.Generator().manual_seed(1)) * 255

    if "f11" in only:
        model = make_ref_model("LarvaNet", argv, seed=0)
        attach_training(model)
        model.volume_per_step = 48 * 48 * 16 * 3
        val = FakeValLoader(7)
        args = types.SimpleNamespace(train_path="/tmp")
        losses, lrs, rec = [], [], {}
        for step in range(3):
            losses.append(model.train_step_larva(args, val, x, truth, None))
            lrs.append(model.get_lr())
            if step == 0:
                for k, p in model.model.named_parameters():
                    gnp = p.grad.detach().numpy()
                    idx = np.random.RandomState(len(k)).choice(gnp.size, min(64, gnp.size), replace=False)
                    rec["gidx." + k] = idx
                    rec["gval." + k] = gnp.ravel()[idx].copy()
                    rec["gmax." + k] = np.array(np.abs(gnp).max())
                    rec["gabs." + k] = np.array(np.abs(gnp.astype(np.float64)).sum())
                    rec["gsha." + k] = np.array(sha(gnp))
        after = sd_to_np(model.model.state_dict())
        flat_after = np.concatenate([after[k].ravel() for k in sorted(after)])
        np.savez(os.path.join(OUT, "f11_m4b4_train_steps.npz"), losses=np.array(losses, np.float64),
                 lrs=np.array(lrs, np.float64), after3_sample=flat_after[::211].copy(),
                 after3_sha=np.array(sha(flat_after)), global_step=np.array(model.global_step),
                 temp_volume=np.array(model.temp_volume), **rec)

    if "f12" in only:
        model = make_ref_model("LarvaNet", argv, seed=0)
        with torch.no_grad():
            y = model.model(x).numpy()
        y8 = np.stack([validate._image_to_uint8(im) for im in y])
        t8 = np.stack([validate._image_to_uint8(im) for im in truth.numpy()])
        psnr = np.array([float(validate._image_psnr(output_image=y8[i], truth_image=t8[i])) for i in range(len(y8))],
                        np.float64)
        np.savez(os.path.join(OUT, "f12_m4b4_uint8.npz"), u8_sha=np.array(sha(y8)), u8_img0=y8[0], u8_img15=y8[15],
                 psnr_vs_truth=psnr, out_img0_f32=y[0].astype(np.float32)[:, ::3, ::3].copy())


def trajectory_batches(nb=4, seed0=1300):
    """The training batches of F13 (2 x 3 x 12 x 12 -> 2 x 3 x 48 x 48), step s uses batch s mod nb."""
    pool = []
    for i in range(nb):
        g = torch.Generator().manual_seed(seed0 + i)
        pool.append((torch.rand(2, 3, 12, 12, generator=g) * 255, torch.rand(2, 3, 48, 48, generator=g) * 255))
    return pool


def synthetic_task(steps=200, batch=4, patch=16, lr_size=40):
    """F16's / F17's data, from the repo's own dataset-free loader (larvanet_amd/dataloaders/synthetic_loader.py: pure
    numpy, seeded): `steps` training batches of batch x 3 x patch x patch (x4 truth) and a 3-image validation loader."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    S = importlib.import_module("larvanet_amd.dataloaders.synthetic_loader")
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


F13_ARGV = ["--num_modules=2", "--num_blocks=2,2", "--lr=2e-3", "--val_volume=1.2e9"]
F13_STEPS = 72
F13_VOLUME_PER_STEP = 400000000   # 3 steps per validation; file names run through vol1G, vol2G, vol4G ...


def main_r3(only):
    """Round-3 fixtures.  F13 = the reference's own train_step_larva driven through its validation branch
    (models/LarvaNet.py:116-137: temp_volume >= val_volume -> total_volume bookkeeping -> validate_for_train ->
    scheduler.step(avg_psnr) at :161 -> save() name at :183-185) for 72 steps at M2B2 on four cycling batches,
    long enough for ReduceLROnPlateau (patience 3, cooldown 6, factor .5, :90-92) to halve the learning rate.
    F14 = LarvaNetV2.train_step_larva (models/LarvaNetV2.py:101-148) at M4B4 on 16x3x48x48 for 3 steps, recorded
    like F11.  `python make_golden.py r3` writes only these."""
    import contextlib
    import io
    import tempfile

    def f13_run(perturb_seed=None):
        model = make_ref_model("LarvaNet", F13_ARGV, seed=0)
        if perturb_seed is not None:
            # the reference's own sensitivity: initial weights moved by about one fp32 ulp (relative 1e-7 * N(0,1))
            gen = torch.Generator().manual_seed(perturb_seed)
            with torch.no_grad():
                for p in model.model.parameters():
                    p.mul_(1 + 1e-7 * torch.randn(p.shape, generator=gen))
        attach_training(model)
        model.volume_per_step = F13_VOLUME_PER_STEP
        val = FakeValLoader(7)
        tmp = tempfile.mkdtemp()
        args = types.SimpleNamespace(train_path=tmp)
        pool = trajectory_batches()
        losses, lrs, vols, val_steps, psnrs = [], [], [], [], []
        for step in range(F13_STEPS):
            x, t = pool[step % len(pool)]
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                losses.append(model.train_step_larva(args, val, x, t, None))
            lrs.append(model.get_lr())
            vols.append((model.total_volume, model.temp_volume))
            for line in buf.getvalue().splitlines():
                if "psnr=" in line:
                    val_steps.append(model.global_step)
                    psnrs.append(float(line.split("psnr=")[1].split(",")[0]))
        return model, tmp, losses, lrs, vols, val_steps, psnrs

    def f13_cleanup(tmp):
        for n in os.listdir(tmp):
            os.remove(os.path.join(tmp, n))
        os.rmdir(tmp)

    if "f13" in only:
        model, tmp, losses, lrs, vols, val_steps, psnrs = f13_run()
        after = sd_to_np(model.model.state_dict())
        flat_after = np.concatenate([after[k].ravel() for k in sorted(after)])
        ckpts = sorted(os.listdir(tmp), key=lambda n: int(n.split("_")[1][4:]))
        last = torch.load(os.path.join(tmp, ckpts[-1]))
        assert all(np.array_equal(last[k].numpy(), after[k]) for k in after)   # the last save is the final weights
        sch = model.scheduler
        f13_cleanup(tmp)
        # How far the REFERENCE moves when its initial weights move by one ulp: this trajectory (lr 2e-3, 72 AdamW
        # steps) amplifies rounding-level differences, so "equal to the reference" can only mean "inside the tube
        # the reference itself sweeps out".  Four perturbed runs; per step / per validation the largest deviation
        # from the unperturbed run, then its running maximum.
        env_loss, env_psnr, env_w, lrs_same = np.zeros(F13_STEPS), np.zeros(len(psnrs)), 0.0, True
        for seed in (11, 12, 13, 14):
            m2, tmp2, l2, lr2, _, _, p2 = f13_run(perturb_seed=seed)
            f13_cleanup(tmp2)
            env_loss = np.maximum(env_loss, np.abs(np.array(l2) / np.array(losses) - 1))
            env_psnr = np.maximum(env_psnr, np.abs(np.array(p2) - np.array(psnrs)))
            a2 = sd_to_np(m2.model.state_dict())
            env_w = max(env_w, float(np.abs(np.concatenate([a2[k].ravel() for k in sorted(a2)]) - flat_after).mean()))
            lrs_same = lrs_same and lr2 == lrs
        np.savez(os.path.join(OUT, "f13_val_trajectory.npz"), losses=np.array(losses, np.float64),
                 lrs=np.array(lrs, np.float64), total_volume=np.array([v[0] for v in vols], np.float64),
                 temp_volume=np.array([v[1] for v in vols], np.float64), val_steps=np.array(val_steps),
                 psnrs=np.array(psnrs, np.float64), ckpt_names=np.array(ckpts),
                 sched_best=np.array(float(sch.best)), sched_num_bad=np.array(int(sch.num_bad_epochs)),
                 sched_cooldown=np.array(int(sch.cooldown_counter)),
                 after_sample=flat_after[::61].copy(), after_sha=np.array(sha(flat_after)),
                 global_step=np.array(model.global_step),
                 ulp_tube_loss=np.maximum.accumulate(env_loss), ulp_tube_psnr=np.maximum.accumulate(env_psnr),
                 ulp_tube_weights_mean=np.array(env_w), ulp_tube_same_lrs=np.array(lrs_same))

    if "f16" in only:
        # A short REALISTIC training run (the stand-in north_star's "PSNR within 0.02 dB of reference" can have without
        # DIV2K): the reference's own train_step_larva for 200 steps at its default learning rate on a learnable task --
        # the repo's dataset-free synthetic loader (smooth images, LR = box-filtered HR), 200 different batches of
        # 4 x 3 x 16 x 16 patches, validation on 3 synthetic images at step 1 and every 25 steps -- plus how far the
        # reference moves when its initial weights move by one ulp (two perturbed runs).
        def f16_run(perturb_seed=None):
            model = make_ref_model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2", "--val_volume=25"], seed=0)
            if perturb_seed is not None:
                gen = torch.Generator().manual_seed(perturb_seed)
                with torch.no_grad():
                    for p_ in model.model.parameters():
                        p_.mul_(1 + 1e-7 * torch.randn(p_.shape, generator=gen))
            attach_training(model)
            model.volume_per_step = 1
            batches, val = synthetic_task()
            tmp = tempfile.mkdtemp()
            args = types.SimpleNamespace(train_path=tmp)
            losses, lrs, psnrs = [], [], []
            for step in range(200):
                x, t = batches[step]
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    losses.append(model.train_step_larva(args, val, x, t, None))
                lrs.append(model.get_lr())
                psnrs += [float(line.split("psnr=")[1].split(",")[0]) for line in buf.getvalue().splitlines() if "psnr=" in line]
            f13_cleanup(tmp)
            return losses, lrs, psnrs

        losses, lrs, psnrs = f16_run()
        tube_l, tube_p = np.zeros(200), np.zeros(len(psnrs))
        for seed in (21, 22):
            l2, _, p2 = f16_run(seed)
            tube_l = np.maximum(tube_l, np.abs(np.array(l2) / np.array(losses) - 1))
            tube_p = np.maximum(tube_p, np.abs(np.array(p2) - np.array(psnrs)))
        np.savez(os.path.join(OUT, "f16_realistic_training.npz"), losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64),
                 psnrs=np.array(psnrs, np.float64), ulp_tube_loss=np.maximum.accumulate(tube_l),
                 ulp_tube_psnr=np.maximum.accumulate(tube_p))

    if "f17" in only:
        # F16's question at the HEADLINE configuration (M4B4, 16 x 3 x 48 x 48 per step: the launch geometry bench.py times):
        # the reference's own train_step_larva for 60 steps on the learnable synthetic task, validation at step 1 and
        # every 20 steps, and one re-run from initial weights one ulp away.
        def f17_run(perturb_seed=None):
            model = make_ref_model("LarvaNet", ["--num_modules=4", "--num_blocks=4,4,4,4", "--val_volume=20"], seed=0)
            if perturb_seed is not None:
                gen = torch.Generator().manual_seed(perturb_seed)
                with torch.no_grad():
                    for p_ in model.model.parameters():
                        p_.mul_(1 + 1e-7 * torch.randn(p_.shape, generator=gen))
            attach_training(model)
            model.volume_per_step = 1
            batches, val = synthetic_task(steps=60, batch=16, patch=48, lr_size=96)
            tmp = tempfile.mkdtemp()
            args = types.SimpleNamespace(train_path=tmp)
            losses, psnrs = [], []
            for step in range(60):
                x, t = batches[step]
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    losses.append(model.train_step_larva(args, val, x, t, None))
                psnrs += [float(line.split("psnr=")[1].split(",")[0]) for line in buf.getvalue().splitlines() if "psnr=" in line]
            f13_cleanup(tmp)
            return losses, psnrs

        losses, psnrs = f17_run()
        l2, p2 = f17_run(31)
        np.savez(os.path.join(OUT, "f17_headline_training.npz"), losses=np.array(losses, np.float64), psnrs=np.array(psnrs, np.float64),
                 ulp_tube_loss=np.maximum.accumulate(np.abs(np.array(l2) / np.array(losses) - 1)),
                 ulp_tube_psnr=np.maximum.accumulate(np.abs(np.array(p2) - np.array(psnrs))))

    if "f15" in only:
        # BASELINE configs[0]: the reference's EDSR through train.py's loop (train.py:83-105: get_next_train_scale,
        # a batch of lists of CHW arrays, model.train_step(input_list, scale, truth_list, summary)), small width here
        edsr = importlib.import_module("models.edsr")
        argv = ["--edsr_conv_features=16", "--edsr_res_blocks=2", "--edsr_learning_rate_decay_steps=2"]
        model = edsr.create_model()
        model.parse_args(argv)
        torch.manual_seed(4)
        model.prepare(is_training=True, scales=[4])
        before = sd_to_np(model.model.state_dict())
        rng = np.random.RandomState(15)
        xs = [rng.randint(0, 256, size=(3, 12, 12)).astype(np.float32) for _ in range(2)]
        ts = [rng.randint(0, 256, size=(3, 48, 48)).astype(np.float32) for _ in range(2)]
        losses, lrs = [], []
        for step in range(4):
            scale = model.get_next_train_scale()
            losses.append(model.train_step(input_list=xs, scale=scale, truth_list=ts, summary=None))
            lrs.append(model.optim.param_groups[0]["lr"])
        after = sd_to_np(model.model.state_dict())
        with torch.no_grad():
            up = model.upscale(input_list=xs[:1], scale=4)[0]
        full = edsr.create_model()
        full.parse_args([])
        torch.manual_seed(4)
        full.prepare(is_training=False, scales=[4])
        full_sd = full.model.state_dict()
        np.savez(os.path.join(OUT, "f15_edsr_train_steps.npz"), x=np.stack(xs), truth=np.stack(ts),
                 losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64), up_sample=up[:, ::3, ::3].copy(),
                 global_step=np.array(model.global_step), default_keys=np.array(sorted(full_sd)),
                 default_params=np.array(sum(v.numel() for v in full_sd.values())),
                 **{"sd." + k: v for k, v in before.items()}, **{"after." + k: v for k, v in after.items()})

    if "f14" in only:
        argv = ["--num_modules=4", "--num_blocks=4,4,4,4"]
        x = torch.rand(16, 3, 48, 48, generator=torch.Generator().manual_seed(0)) * 255
        truth = torch.rand(16, 3, 192, 192, generator=torch.Generator().manual_seed(1)) * 255
        model = make_ref_model("LarvaNetV2", argv, seed=3)
        attach_training(model)
        model.volume_per_step = 48 * 48 * 16 * 3
        val = FakeValLoader(7)
        args = types.SimpleNamespace(train_path="/tmp")
        losses, lrs, rec = [], [], {}
        before = sd_to_np(model.model.state_dict())
        flat_before = np.concatenate([before[k].ravel() for k in sorted(before)])
        for step in range(3):
            losses.append(model.train_step_larva(args, val, x, truth, None))
            lrs.append(model.get_lr())
            if step == 0:
                for k, p in model.model.named_parameters():
                    gnp = p.grad.detach().numpy()
                    idx = np.random.RandomState(len(k)).choice(gnp.size, min(64, gnp.size), replace=False)
                    rec["gidx." + k] = idx
                    rec["gval." + k] = gnp.ravel()[idx].copy()
                    rec["gmax." + k] = np.array(np.abs(gnp).max())
                    rec["gabs." + k] = np.array(np.abs(gnp.astype(np.float64)).sum())
                    rec["gsha." + k] = np.array(sha(gnp))
        after = sd_to_np(model.model.state_dict())
        flat_after = np.concatenate([after[k].ravel() for k in sorted(after)])
        # The same first step by the reference in float64 (same initial weights): how far the reference's OWN fp32
        # gradients are from the exact ones -- the L1 gradient is sign(out - truth), a handful of the 8.8 M pairs of
        # the five exits sit within the forward's rounding error of each other, and a flipped sign moves single
        # weight-gradient elements by a few 1e-4 of the tensor's maximum.
        model64 = make_ref_model("LarvaNetV2", argv, seed=3)
        model64.model.double()
        attach_training(model64)
        model64.volume_per_step = 48 * 48 * 16 * 3
        model64.global_step = 1   # (not step 1: its validation would feed float32 images to the float64 module)
        model64.train_step_larva(args, val, x.double(), truth.double(), None)
        ref32_vs_64 = 0.0
        for k, p in model64.model.named_parameters():
            g64 = p.grad.detach().numpy().ravel()[rec["gidx." + k]]
            rec["gval64." + k] = g64.copy()
            ref32_vs_64 = max(ref32_vs_64, float(np.abs(g64 - rec["gval." + k]).max() / max(float(rec["gmax." + k]), 1e-30)))
        rec["ref32_vs_ref64_worst"] = np.array(ref32_vs_64)
        np.savez(os.path.join(OUT, "f14_v2_m4b4_train_steps.npz"), losses=np.array(losses, np.float64),
                 lrs=np.array(lrs, np.float64), before_sha=np.array(sha(flat_before)), before_sample=flat_before[::211].copy(),
                 after3_sample=flat_after[::211].copy(), after3_sha=np.array(sha(flat_after)),
                 global_step=np.array(model.global_step), temp_volume=np.array(model.temp_volume), **rec)


def main():
    sys.path.insert(0, REF)
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(False)
    if len(sys.argv) > 1 and sys.argv[1] == "r2":
        main_r2(sys.argv[2:] or ["f11", "f12"])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "r3":
        main_r3(sys.argv[2:] or ["f13", "f14", "f15", "f16", "f17"])
        return

    # ---------------- F1/F2: M2B2 weights, staged outputs on a 2x3x12x12 input ----------------
    model = make_ref_model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], seed=0)
    net = model.model
    sd = sd_to_np(net.state_dict())
    g = torch.Generator().manual_seed(100)
    x = torch.rand(2, 3, 12, 12, generator=g) * 255
    stages = {}

    def hook(name):
        def fn(_m, _i, o):
            stages[name] = o.detach().numpy().copy()
        return fn

    handles = [net.head.register_forward_hook(hook("head")),
               net.body_0.register_forward_hook(hook("body_0")),
               net.body_1.register_forward_hook(hook("body_1")),
               net.body_0.res_blocks[0].register_forward_hook(hook("body_0.res_blocks.0")),
               net.body_0.res_blocks[0].body[1].register_forward_hook(hook("body_0.res_blocks.0.relu")),
               net.body_0.leg.recon_block.register_forward_hook(hook("body_0.leg.recon_block"))]
    with torch.no_grad():
        fea = net.head(x)
        base = net.base(x)
        outs = []
        for i in range(2):
            fea = getattr(net, "body_%d" % i)(fea)
            outs.append(getattr(net, "body_%d" % i).leg(fea, base).numpy().copy())
        final = net(x).numpy().copy()
    for h in handles:
        h.remove()
    np.savez(os.path.join(OUT, "f1_m2b2_forward.npz"), x=x.numpy(), base=base.numpy(), exit_0=outs[0],
             exit_1=outs[1], final=final, **{"stage." + k: v for k, v in stages.items()},
             **{"sd." + k: v for k, v in sd.items()})

    # ---------------- F3: PixelShuffle(4) integer KAT (bit-exact index layout) ----------------
    ps_in = torch.arange(2 * 48 * 3 * 5, dtype=torch.int32).reshape(2, 48, 3, 5)
    ps_out = torch.nn.PixelShuffle(4)(ps_in)
    np.savez(os.path.join(OUT, "f3_pixel_shuffle.npz"), inp=ps_in.numpy(), out=ps_out.numpy())

    # ---------------- F4: bicubic x4 incl. borders ----------------
    g = torch.Generator().manual_seed(101)
    bx = torch.rand(1, 3, 9, 11, generator=g) * 255
    np.savez(os.path.join(OUT, "f4_bicubic.npz"), inp=bx.numpy(), out=net.base(bx).numpy())

    # ---------------- F5: the reference's own train_step_larva, 3 steps ----------------
    model = make_ref_model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], seed=0)
    attach_training(model)
    model.volume_per_step = 12 * 12 * 2 * 3
    g = torch.Generator().manual_seed(102)
    tx = torch.rand(2, 3, 12, 12, generator=g) * 255
    tt = torch.rand(2, 3, 48, 48, generator=g) * 255
    val = FakeValLoader(7)
    args = types.SimpleNamespace(train_path="/tmp")
    losses, grads1, lrs = [], None, []
    for step in range(3):
        losses.append(model.train_step_larva(args, val, tx, tt, None))
        lrs.append(model.get_lr())
        if step == 0:
            grads1 = {k: p.grad.detach().numpy().copy() for k, p in model.model.named_parameters()}
    after = sd_to_np(model.model.state_dict())
    flat_after = np.concatenate([after[k].ravel() for k in sorted(after)])
    np.savez(os.path.join(OUT, "f5_train_steps.npz"), x=tx.numpy(), truth=tt.numpy(),
             losses=np.array(losses, np.float64), lrs=np.array(lrs, np.float64),
             after3_sample=flat_after[::61].copy(), after3_sha=np.array(sha(flat_after)),
             global_step=np.array(model.global_step), temp_volume=np.array(model.temp_volume),
             **{"grad1." + k: v for k, v in grads1.items()})

    # ---------------- F6: canonical M4B4 forward on 16x3x48x48 (sampled) ----------------
    model = make_ref_model("LarvaNet", ["--num_modules=4", "--num_blocks=4,4,4,4"], seed=0)
    sd4 = sd_to_np(model.model.state_dict())
    flat4 = np.concatenate([sd4[k].ravel() for k in sorted(sd4)])
    g = torch.Generator().manual_seed(0)
    cx = torch.rand(16, 3, 48, 48, generator=g) * 255
    with torch.no_grad():
        cy = model.model(cx).numpy()
    idx = np.random.RandomState(5).choice(cy.size, 4096, replace=False)
    np.savez(os.path.join(OUT, "f6_m4b4_canonical.npz"), sd_sha=np.array(sha(flat4)),
             sd_sample=flat4[::997].copy(), n_params=np.array(flat4.size), out_sha=np.array(sha(cy)),
             sample_idx=idx, sample_val=cy.ravel()[idx], out_mean=np.array(cy.mean(dtype=np.float64)))

    # ---------------- F7: validate.py helpers ----------------
    validate = importlib.import_module("validate")
    img = np.array([[[-3.2, 0.5, 1.5, 2.5], [254.5, 255.5, 300.0, 127.49999]],
                    [[0.49999997, 3.5, 4.5, 99.5], [1e-7, -0.5, 255.49998, 128.5]],
                    [[10.2, 20.7, 30.5, 31.5], [32.5, 33.5, 250.5, 251.5]]], np.float32)
    u8 = validate._image_to_uint8(img)
    rng = np.random.RandomState(3)
    o_img = rng.randint(0, 256, size=(3, 8, 10)).astype(np.uint8)
    t_img = np.clip(o_img.astype(np.int32) + rng.randint(-9, 10, size=o_img.shape), 0, 255).astype(np.uint8)
    t_big = rng.randint(0, 256, size=(3, 11, 13)).astype(np.uint8)
    t_big[:, :8, :10] = t_img
    fitted = validate._fit_truth_image_size(output_image=o_img, truth_image=t_big)
    psnr = validate._image_psnr(output_image=o_img, truth_image=fitted)
    np.savez(os.path.join(OUT, "f7_validate_helpers.npz"), img=img, u8=u8, o_img=o_img, t_big=t_big,
             fitted=fitted, psnr=np.array(float(psnr), np.float64))

    # ---------------- F8: LarvaNetV2 (tail) forward + one train step ----------------
    model2 = make_ref_model("LarvaNetV2", ["--num_modules=2", "--num_blocks=2,2"], seed=0)
    sd2 = sd_to_np(model2.model.state_dict())
    same_trunk = all(np.array_equal(sd2[k], sd[k]) for k in sd)
    g = torch.Generator().manual_seed(103)
    vx = torch.rand(2, 3, 12, 12, generator=g) * 255
    vt = torch.rand(2, 3, 48, 48, generator=g) * 255
    with torch.no_grad():
        v_out = model2.model(vx).numpy().copy()
    attach_training(model2)
    model2.volume_per_step = 0
    model2.steps_per_epoch = 10 ** 9 if hasattr(model2, "steps_per_epoch") else None
    v_loss = model2.train_step_larva(types.SimpleNamespace(train_path="/tmp"), FakeValLoader(7), vx, vt, None)
    v_grads = {k: p.grad.detach().numpy().copy() for k, p in model2.model.named_parameters() if k.startswith("tail.")}
    trunk_gsum = {k: float(p.grad.double().abs().sum()) for k, p in model2.model.named_parameters()
                  if not k.startswith("tail.")}
    np.savez(os.path.join(OUT, "f8_v2_tail.npz"), x=vx.numpy(), truth=vt.numpy(), out=v_out,
             loss=np.array(float(v_loss), np.float64), same_trunk=np.array(same_trunk),
             trunk_keys=np.array(sorted(trunk_gsum)), trunk_gabs=np.array([trunk_gsum[k] for k in sorted(trunk_gsum)]),
             **{"sd." + k: v for k, v in sd2.items() if k.startswith("tail.")},
             **{"grad." + k: v for k, v in v_grads.items()})

    # ---------------- F9: chop-forward split / stitch on an odd-sized image ----------------
    image_utils = importlib.import_module("utils.image_utils")

    class NearestModel:
        def upscale(self, input_list, scale):
            a = np.asarray(input_list, np.float32)
            return a.repeat(scale, axis=2).repeat(scale, axis=3) + 0.25

    rng = np.random.RandomState(11)
    cimg = rng.randint(0, 256, size=(3, 21, 27)).astype(np.float32)
    splits = image_utils._split_image(cimg, chop=True, overlap_size=6)
    chopped = image_utils.upscale_with_chop_forward(model=NearestModel(), input_image=cimg, scale=4, overlap_size=6)
    np.savez(os.path.join(OUT, "f9_chop_forward.npz"), img=cimg, out=chopped,
             split_shapes=np.array([s.shape for s in splits]), **{"split%d" % i: s for i, s in enumerate(splits)})

    # ---------------- F10: LarvaNet.upscale + PSNR protocol on a small image ----------------
    model = make_ref_model("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"], seed=0)
    rng = np.random.RandomState(21)
    lr_img = rng.randint(0, 256, size=(3, 20, 26)).astype(np.float32)
    hr_img = rng.randint(0, 256, size=(3, 80, 104)).astype(np.float32)
    with torch.no_grad():
        up = model.upscale(input_list=[lr_img], scale=4)[0]
    o8 = validate._image_to_uint8(up)
    t8 = validate._fit_truth_image_size(output_image=o8, truth_image=validate._image_to_uint8(hr_img))
    np.savez(os.path.join(OUT, "f10_upscale_psnr.npz"), lr=lr_img, hr=hr_img, up=up,
             psnr=np.array(float(validate._image_psnr(output_image=o8, truth_image=t8)), np.float64))

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
