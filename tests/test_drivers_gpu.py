"""End-to-end drivers on the GPU: a few training steps with validation + checkpoint, then the
validation driver (plain and chop-forward) on that checkpoint."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_then_validate(hip_device, tmp_path, capsys):
    from larvanet_amd import train_larva, validate
    vol = 12 * 12 * 4 * 3
    model = train_larva.main([
        "--model=LarvaNet", "--dataloader=synthetic_loader", "--val_dataloader=synthetic_loader",
        "--train_path", str(tmp_path), "--max_steps=5", "--batch_size=4", "--input_patch_size=12",
        "--num_modules=2", "--num_blocks=1,1", "--synthetic_images=3", "--synthetic_lr_size=20",
        "--val_volume=%d" % (2 * vol)])
    out = capsys.readouterr().out
    assert model.global_step == 5
    assert out.count("begin validation") == 3  # step 1, then every 2 steps of volume (steps 2 and 4)
    ckpts = sorted(glob.glob(os.path.join(str(tmp_path), "model_step*_vol*.pth")))
    assert len(ckpts) == 2 and os.path.basename(ckpts[0]).startswith("model_step2_")
    common = ["--model=LarvaNet", "--dataloader=synthetic_loader", "--num_modules=2", "--num_blocks=1,1",
              "--synthetic_images=3", "--synthetic_lr_size=20", "--synthetic_uint8", "--restore_path", ckpts[-1]]
    plain = validate.main(common + ["--save_path", str(tmp_path / "sr")])
    assert np.isfinite(plain[4]["psnr"]) and len(plain[4]["per_image"]) == 3
    assert len(glob.glob(os.path.join(str(tmp_path), "sr", "x4", "*.png"))) == 3
    chop = validate.main(common + ["--chop_forward", "--chop_overlap_size=8"])
    assert np.isfinite(chop[4]["psnr"]) and abs(chop[4]["psnr"] - plain[4]["psnr"]) < 1.0


def test_validate_scores_on_the_device_like_on_the_host(hip_device):
    """validate.py without --save_path scores every image with one kernel on the device (8 bytes back
    instead of the HR image); --host_psnr is the reference's numpy protocol: same PSNR to 1e-4 dB."""
    from larvanet_amd import validate
    common = ["--model=LarvaNet", "--dataloader=synthetic_loader", "--num_modules=2", "--num_blocks=1,1",
              "--synthetic_images=3", "--synthetic_lr_size=21", "--synthetic_uint8"]
    torch.manual_seed(0)
    dev = validate.main(list(common))
    torch.manual_seed(0)
    host = validate.main(common + ["--host_psnr"])
    a, b = dev[4]["per_image"], host[4]["per_image"]
    assert [r[0] for r in a] == [r[0] for r in b] == [0, 1, 2]
    # (exact integer sum on the device, float32 mean on the host)
    assert all(abs(x[1] - y[1]) < 1e-4 for x, y in zip(a, b))


def test_train_larva_v2_driver(hip_device, tmp_path, capsys):
    """train_larvaV2.py's loop: steps_per_epoch on the model, validation at step 1 only (the reference never sets
    volume_per_step in this driver), no checkpoint."""
    from larvanet_amd import train_larvaV2
    model = train_larvaV2.main([
        "--model=LarvaNetV2", "--dataloader=synthetic_loader", "--val_dataloader=synthetic_loader",
        "--train_path", str(tmp_path), "--max_steps=4", "--batch_size=2", "--input_patch_size=12",
        "--num_modules=2", "--num_blocks=1,1", "--synthetic_images=2", "--synthetic_lr_size=16", "--steps_per_epoch=3",
        "--val_volume=1"])
    out = capsys.readouterr().out
    assert model.global_step == 4 and model.steps_per_epoch == 3 and model.volume_per_step == 0
    assert "3 steps equal to 1 epoch" in out and out.count("begin validation") == 1
    assert not glob.glob(os.path.join(str(tmp_path), "model_step*"))
