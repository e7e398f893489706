"""Host-side logic of the plugin surface, loaders and drivers (CPU only, no kernels run)."""
import json
import os
import types

import numpy as np
import pytest
import torch


def _make(name, argv, training=True):
    import importlib
    m = importlib.import_module("larvanet_amd.models." + name).create_model()
    parsed, rest = m.parse_args(argv)
    return m, parsed, rest


def test_parse_args_chaining_returns_copy_and_leftovers():
    m, parsed, rest = _make("LarvaNet", ["--num_modules=4", "--num_blocks=4,4,4,4", "--foo=1", "--lr=1e-3"])
    assert rest == ["--foo=1"] and parsed.lr == 1e-3 and parsed.num_modules == 4
    parsed.lr = 5.0
    assert m.args.lr == 1e-3  # deep copy, like the reference (models/LarvaNet.py:65-66)
    assert m.args.val_volume == 30e9 and m.args.cooldown == 6 and m.args.min_lr == 1e-8
    m2, p2, _ = _make("LarvaNetV2", ["--num_modules=2", "--num_blocks=1,1"])
    assert p2.val_volume == 3e9 and p2.lr == 1e-4 and p2.min_lr == 1e-7 and not hasattr(p2, "cooldown")


def test_unsupported_interpolate_mode_is_refused_when_the_flags_are_parsed():
    """The reference hands any --interpolate string to F.interpolate(..., align_corners=False)
    (models/LarvaNet.py:283-285): bicubic and bilinear work there and have HIP kernels; every other mode raises
    in that call (checked here against F.interpolate itself) and is refused when the flags are parsed."""
    import importlib
    import torch
    import torch.nn.functional as F
    m = importlib.import_module("larvanet_amd.models.LarvaNet").create_model()
    for bad in ("nearest", "nearest-exact", "area", "trilinear", "linear", "lanczos"):
        with pytest.raises(ValueError, match="bicubic"):
            m.parse_args(["--num_modules=1", "--num_blocks=1", "--interpolate=" + bad])
        with pytest.raises((NotImplementedError, ValueError)):
            F.interpolate(torch.zeros(1, 3, 4, 4), scale_factor=4, mode=bad, align_corners=False)
    for ok in ("bicubic", "bilinear"):
        m.parse_args(["--num_modules=1", "--num_blocks=1", "--interpolate=" + ok])
        F.interpolate(torch.zeros(1, 3, 4, 4), scale_factor=4, mode=ok, align_corners=False)


def test_prepare_validates_scales_and_block_list():
    m, _, _ = _make("LarvaNet", ["--num_modules=2", "--num_blocks=2,2"])
    with pytest.raises(ValueError):
        m.prepare(is_training=False, scales=[5])
    with pytest.raises(ValueError):
        m.prepare(is_training=False, scales=[2, 4])
    bad, _, _ = _make("LarvaNet", ["--num_modules=3", "--num_blocks=2,2"])
    with pytest.raises(GeneratorExit):
        bad.prepare(is_training=False, scales=[4])


def test_state_dict_keys_are_the_reference_wire_format():
    m, _, _ = _make("LarvaNet", ["--num_modules=4", "--num_blocks=4,4,4,4"])
    m.prepare(is_training=True, scales=[4])
    sd = m.model.state_dict()
    assert len(sd) == 82 and sum(v.numel() for v in sd.values()) == 832704
    assert "head.feature_extraction.weight" in sd and "body_3.res_blocks.3.body.2.bias" in sd
    assert "body_0.leg.recon_block.0.weight" in sd and tuple(sd["head.feature_extraction.weight"].shape) == (48, 3, 3, 3)
    assert isinstance(m.optim, torch.optim.AdamW) and m.get_lr() == 4e-4
    assert m.optim.defaults["weight_decay"] == 0.01 and m.optim.defaults["betas"] == (0.9, 0.999)
    assert m.get_next_train_scale() == 4 and m.get_model() is m.model
    v2, _, _ = _make("LarvaNetV2", ["--num_modules=4", "--num_blocks=4,4,4,4"])
    v2.prepare(is_training=False, scales=[4])
    sd2 = v2.model.state_dict()
    assert len(sd2) == 88 and sum(v.numel() for v in sd2.values()) == 957264
    # models/LarvaLegV2.py: V2's parameters and flag defaults (lr 1e-4, val_volume 3e9) plus --leg (default 4)
    lv2, args, _ = _make("LarvaLegV2", ["--num_modules=4", "--num_blocks=4,4,4,4"])
    lv2.prepare(is_training=False, scales=[4])
    assert sorted(lv2.model.state_dict()) == sorted(sd2) and (args.leg, args.lr, args.val_volume) == (4, 1e-4, 3e9)
    with pytest.raises(ValueError):
        _make("LarvaLegV2", ["--num_modules=2", "--num_blocks=1,1", "--leg=3"])[0].prepare(is_training=False, scales=[4])
    assert tuple(sd2["tail.merge_conv.weight"].shape) == (48, 192, 3, 3)


def test_scheduler_is_plateau_on_max_psnr():
    m, _, _ = _make("LarvaNet", ["--num_modules=1", "--num_blocks=1", "--patience=0", "--cooldown=0"])
    m.prepare(is_training=True, scales=[4])
    m.scheduler.step(30.0)
    m.scheduler.step(30.0005)  # below the absolute threshold of 1e-3 dB: counts as no improvement
    assert m.get_lr() == pytest.approx(2e-4)


def test_cpu_forward_fails_loudly():
    m, _, _ = _make("LarvaNet", ["--num_modules=1", "--num_blocks=1"])
    m.prepare(is_training=False, scales=[4])
    if m.device.type == "cpu":
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.upscale([np.zeros((3, 8, 8), np.float32)], 4)


def test_metrics_and_chop_forward_match_reference_vectors(golden):
    from larvanet_amd import image_utils, metrics
    g = golden("f7_validate_helpers.npz")
    assert np.array_equal(metrics.image_to_uint8(g["img"]), g["u8"])
    g9 = golden("f9_chop_forward.npz")
    parts = image_utils.split_quadrants(g9["img"], 6)
    for i, p in enumerate(parts):
        assert np.array_equal(p, g9["split%d" % i])

    class Nearest:
        def upscale(self, input_list, scale):
            a = np.asarray(input_list, np.float32)
            return a.repeat(scale, axis=2).repeat(scale, axis=3) + 0.25

    assert np.array_equal(image_utils.upscale_with_chop_forward(Nearest(), g9["img"], 4, 6), g9["out"])


def _write_div2k(tmp_path, n=3):
    from PIL import Image
    from larvanet_amd.dataloaders.synthetic_loader import make_pair
    hr_dir = tmp_path / "HR"
    lr_dir = tmp_path / "LR" / "X4"
    hr_dir.mkdir(parents=True)
    lr_dir.mkdir(parents=True)
    for i in range(n):
        lr, hr = make_pair(i, 20 + i, 24, 4, False)
        Image.fromarray(hr.transpose(1, 2, 0)).save(hr_dir / ("%04d.png" % i))
        Image.fromarray(lr.transpose(1, 2, 0)).save(lr_dir / ("%04dx4.png" % i))
    return str(tmp_path / "LR"), str(hr_dir)


def test_div2k_loader_patch_geometry(tmp_path):
    from larvanet_amd.dataloaders import div2k_train_loader, div2k_val_loader
    lr_root, hr_root = _write_div2k(tmp_path)
    ld = div2k_train_loader.create_loader()
    _, rest = ld.parse_args(["--data_input_path", lr_root, "--data_truth_path", hr_root, "--data_seed=3", "--x=1"])
    assert rest == ["--x=1"]
    ld.prepare([4])
    assert ld.get_num_images() == 3 and not ld.is_threaded
    lr, hr, name = ld.get_image_pair(1, 4)
    assert lr.dtype == np.float32 and lr.shape == (3, 21, 24) and hr.shape == (3, 84, 96) and name == "0001"
    ins, outs = ld.get_patch_batch(batch_size=5, scale=4, input_patch_size=8)
    assert len(ins) == 5 and all(a.shape == (3, 8, 8) for a in ins) and all(b.shape == (3, 32, 32) for b in outs)
    # the HR patch is the x4 region of the LR patch under the same rotation/flip: its 4x4 box
    # mean equals the LR patch (the synthetic LR is the rounded box filter of HR)
    for a, b in zip(ins, outs):
        box = np.asarray(b).reshape(3, 8, 4, 8, 4).mean(axis=(2, 4))
        assert np.abs(box - np.asarray(a)).max() <= 0.5 + 1e-4
    # same seed -> same stream; different rank seed -> different stream
    ld2 = div2k_train_loader.create_loader()
    ld2.parse_args(["--data_input_path", lr_root, "--data_truth_path", hr_root, "--data_seed=3"])
    ld2.prepare([4])
    ins2, _ = ld2.get_patch_batch(batch_size=5, scale=4, input_patch_size=8)
    assert all(np.array_equal(a, b) for a, b in zip(ins, ins2))
    val = div2k_val_loader.create_loader()
    val.parse_args(["--val_input_path", lr_root, "--val_truth_path", hr_root])
    val.prepare([4])
    vlr, vhr, _ = val.get_image_pair(0, 4)
    assert vlr.dtype == np.uint8 and vhr.dtype == np.uint8


def test_combined_loader_threads(tmp_path):
    from larvanet_amd.dataloaders import combined_loader
    lr_root, hr_root = _write_div2k(tmp_path)
    ld = combined_loader.create_loader()
    ld.parse_args(["--data_input_path", lr_root, "--data_truth_path", hr_root, "--data_cached",
                   "--data_num_queue_runners=2", "--data_seed=1"])
    ld.prepare([4])
    assert ld.is_threaded and ld.get_queue_data(4) is None
    ld.start_training_queue_runner(batch_size=3, input_patch_size=8)
    for _ in range(4):
        ins, outs = ld.get_queue_data(scale=4)
        assert len(ins) == 3 and ins[0].shape == (3, 8, 8) and outs[0].shape == (3, 32, 32)
    ld.stop_queue_runners()
    assert ld.get_queue_data(4) is None


def test_synthetic_loader_rank_streams_differ(monkeypatch):
    from larvanet_amd import dist as ldist
    from larvanet_amd.dataloaders import synthetic_loader
    batches = []
    for r in (0, 1):
        monkeypatch.setattr(ldist, "rank", lambda r=r: r)
        ld = synthetic_loader.create_loader()
        ld.parse_args(["--synthetic_images=4", "--synthetic_lr_size=24"])
        ld.prepare([4])
        batches.append(ld.get_patch_batch(4, 4, 12)[0])
    assert not all(np.array_equal(a, b) for a, b in zip(*batches))


def test_train_driver_argument_chain_on_cpu(tmp_path, monkeypatch, capsys):
    """Everything of train_larva.main() up to the first step, with the model's device work stubbed
    out: flags are chained driver -> loader -> val loader -> model, arguments.json is written,
    volume_per_step uses the global batch."""
    from larvanet_amd import train_larva
    from larvanet_amd.models import LarvaNet as L
    calls = {}

    def fake_step(self, args, val_dataloader, input_tensor, truth_tensor, summary=None):
        self.global_step += 1
        calls.setdefault("shapes", []).append((tuple(input_tensor.shape), tuple(truth_tensor.shape), input_tensor.dtype))
        return 1.0

    monkeypatch.setattr(L.LarvaNet, "train_step_larva", fake_step)
    model = train_larva.main(["--model=LarvaNet", "--dataloader=synthetic_loader", "--val_dataloader=synthetic_loader",
                              "--train_path", str(tmp_path), "--max_steps=2", "--batch_size=3", "--input_patch_size=12",
                              "--num_modules=1", "--num_blocks=1", "--synthetic_images=2", "--synthetic_lr_size=24",
                              "--bogus_flag=7"])
    assert model.global_step == 2 and model.volume_per_step == 12 * 12 * 3 * 3
    assert calls["shapes"][0] == ((3, 3, 12, 12), (3, 3, 48, 48), torch.float32)
    saved = json.load(open(os.path.join(str(tmp_path), "arguments.json")))
    assert saved["model"] == "LarvaNet" and saved["num_blocks"] == "1" and saved["batch_size"] == 3 and saved["world_size"] == 1
    assert "WARNING: found unhandled arguments: ['--bogus_flag=7']" in capsys.readouterr().out


def test_train_larva_v2_driver_flags():
    """larvanet_amd/train_larvaV2.py: the reference's --steps_per_epoch (train_larvaV2.py:29,73-81) and its one-significant-
    digit default epoch."""
    from larvanet_amd import train_larva
    args, rest = train_larva.build_parser(v2=True).parse_known_args(["--steps_per_epoch=500", "--model=LarvaNetV2", "--x=1"])
    assert args.steps_per_epoch == 500 and rest == ["--x=1"]
    assert not hasattr(train_larva.build_parser().parse_known_args([])[0], "steps_per_epoch")
    assert train_larva.round_to_1(300 * 1024 ** 2 / (48 ** 2 * 16 * 3)) == 3000.0   # 2844.4 -> one significant digit
    assert train_larva.round_to_1(0.0234) == 0.02


def test_one_rule_decides_eager_inference_and_the_direct_head_kernel(monkeypatch):
    """`_infer` (capture or eager) and HeadFn (MFMA or direct head kernel) used two differently computed
    thresholds; both now ask autograd.is_large_inference(N, H, W) on the unpadded input.  An unknown LARVA_HEAD_DIRECT
    value is refused instead of silently meaning `auto`."""
    import inspect
    from larvanet_amd import autograd as A
    from larvanet_amd.models import LarvaNet as L
    assert A.is_large_inference(1, 339, 510) and not A.is_large_inference(16, 48, 48)
    assert not A.is_large_inference(1, 250, 400) and A.is_large_inference(1, 250, 401)
    assert "is_large_inference" in inspect.getsource(L.LarvaNet._infer)
    assert "is_large_inference" in inspect.getsource(A.HeadFn.forward)
    for ok, want in (("auto", "auto"), ("0", False), ("1", True)):
        monkeypatch.setenv("LARVA_HEAD_DIRECT", ok)
        assert A._head_direct_setting() == want
    monkeypatch.setenv("LARVA_HEAD_DIRECT", "yes")
    with pytest.raises(RuntimeError, match="LARVA_HEAD_DIRECT"):
        A._head_direct_setting()


def test_ring_wait_checker_compares_against_the_constants_the_kernel_emits(tmp_path):
    """tools/check_aux_loads.py (run by larvanet_amd/build.py on every build): a counted `s_waitcnt vmcnt(N)` of the LDS-DMA
    ring is checked against the `; LARVA_RING pieces=P aux=A` comment the kernel emits in front of it -- exact values, not
    plausible ranges.  A role whose compiler-emitted operand loads fall short of A, a wait that is not P + A
    and a loader wait that is not (ahead - 1) x pieces are all refused."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "check_aux_loads.py")

    def listing(first_n, loads, later_n, loader_n):
        body = ["_ZN5larva9fake_kernEv:"]
        body += ["\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds"] * 5
        body += ["\tbuffer_load_dwordx4 v[2:5], v6, s[0:3], 0 offen"] * loads
        body += ["\t; LARVA_RING pieces=5 aux=3", "\ts_waitcnt vmcnt(%d)" % first_n, "\ts_barrier"]
        body += ["\t; LARVA_RING pieces=0 aux=3", "\ts_waitcnt vmcnt(%d)" % later_n, "\ts_barrier"]
        body += ["\t; LARVA_RING loader chunk_pieces=20 ahead=2", "\ts_waitcnt vmcnt(%d)" % loader_n, "\ts_barrier"]
        body += [".Lfunc_end0:"]
        return "\n".join(body) + "\n"

    def run(text):
        p = tmp_path / "k.s"
        p.write_text(text)
        return subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)

    ok = run(listing(8, 3, 3, 20))
    assert ok.returncode == 0 and "3 ring waits checked" in ok.stdout, ok.stdout
    for bad in (listing(8, 2, 3, 20),     # the compiler emitted one operand load fewer than the constant counts
                listing(7, 3, 3, 20),     # the first wait keeps fewer than pieces + aux in flight
                listing(8, 3, 2, 20),     # a later wait
                listing(8, 3, 3, 12)):    # the loader's
        r = run(bad)
        assert r.returncode == 1 and "MISMATCH" in r.stdout, r.stdout
