"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.

The LarvaNet graph written with the torch CPU operators the reference itself calls
(F.conv2d / relu / pixel_shuffle / interpolate / l1_loss), functional style over a state_dict
with the reference's key names, with autograd.  It exists (a) to check whole-network outputs
and gradients of the HIP path at sizes the C restatement would take minutes for, and (b) as the
"reference CPU path" that bench.py times on the host cores (cpu_baseline, kind "port").
Pinned against the imported reference by tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F


def head(sd, x):
    """models/LarvaNet.py:223-233"""
    return F.conv2d(x, sd["head.feature_extraction.weight"], sd["head.feature_extraction.bias"], padding=1)


def base_image(x, mode="bicubic"):
    """models/LarvaNet.py:283-285"""
    return F.interpolate(x, scale_factor=4, mode=mode, align_corners=False)


def residual_block(sd, prefix, x):
    """models/LarvaNet.py:205-220"""
    h = F.relu(F.conv2d(x, sd[prefix + ".body.0.weight"], sd[prefix + ".body.0.bias"], padding=1))
    return torch.add(x, F.conv2d(h, sd[prefix + ".body.2.weight"], sd[prefix + ".body.2.bias"], padding=1))


def body(sd, i, x, num_blocks):
    """models/LarvaNet.py:236-248"""
    fea = x
    for j in range(num_blocks):
        fea = residual_block(sd, "body_%d.res_blocks.%d" % (i, j), fea)
    return x + fea


def leg(sd, prefix, fea, base):
    """models/LarvaNet.py:251-267"""
    h = F.relu(F.conv2d(fea, sd[prefix + ".recon_block.0.weight"], sd[prefix + ".recon_block.0.bias"], padding=1))
    out = F.pixel_shuffle(F.conv2d(h, sd[prefix + ".recon_block.2.weight"], sd[prefix + ".recon_block.2.bias"],
                                   padding=1), 4)
    return out + base


def forward_exits(sd, x, blocks, mode="bicubic"):
    """models/LarvaNet.py:102-108 (mode = --interpolate, models/LarvaNet.py:57)"""
    fea = head(sd, x)
    base = base_image(x, mode)
    outs, feats = [], []
    for i, nb in enumerate(blocks):
        fea = body(sd, i, fea, nb)
        feats.append(fea)
        outs.append(leg(sd, "body_%d.leg" % i, fea, base))
    return outs, feats, base


def forward(sd, x, blocks, mode="bicubic"):
    """models/LarvaNet.py:287-293"""
    fea = head(sd, x)
    for i, nb in enumerate(blocks):
        fea = body(sd, i, fea, nb)
    return leg(sd, "body_%d.leg" % (len(blocks) - 1), fea, base_image(x, mode))


def tail(sd, feats, base):
    """models/LarvaNetV2.py:314-334"""
    fea = F.conv2d(torch.cat(feats, dim=1), sd["tail.merge_conv.weight"], sd["tail.merge_conv.bias"], padding=1)
    return leg(sd, "tail", fea, base)


def forward_v2(sd, x, blocks, mode="bicubic"):
    """models/LarvaNetV2.py:355-365"""
    _, feats, base = forward_exits(sd, x, blocks, mode)
    return tail(sd, feats, base)


def multi_exit_loss(sd, x, truth, blocks, v2=False, mode="bicubic"):
    """models/LarvaNet.py:104-109 / models/LarvaNetV2.py:108-123"""
    outs, feats, base = forward_exits(sd, x, blocks, mode)
    loss = 0
    for o in outs:
        loss = loss + F.l1_loss(o, truth)
    if v2:
        loss = loss + F.l1_loss(tail(sd, feats, base), truth)
        return loss / (len(blocks) + 1)
    return loss / len(blocks)


def init_state_dict(blocks, v2=False, seed=None, nf=48):
    """nf: the width of the head / bodies / first leg convs (--num_filters; the reference's is 48, the legs' last conv
    always has 48 = 3 * 4**2 outputs).  Reference initialisation (models/LarvaNet.py:22-31,215,229,260): kaiming_normal_(fan_in,
    a=0) * 0.1, zero bias, drawn in module-construction order (head, then per body: blocks'
    conv1, conv2 ..., then the leg's two convs; V2: tail.merge_conv is constructed first but
    initialised after tail.recon_block, models/LarvaNetV2.py:317-324)."""
    if seed is not None:
        torch.manual_seed(seed)
    sd = {}

    def conv(prefix, cout, cin):
        # nn.Conv2d's own default init consumes RNG first (kaiming_uniform weight, uniform bias)
        w = torch.empty(cout, cin, 3, 3)
        torch.nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        b = torch.empty(cout)
        bound = 1.0 / (cin * 9) ** 0.5
        torch.nn.init.uniform_(b, -bound, bound)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = w, b

    def reinit(prefix):
        torch.nn.init.kaiming_normal_(sd[prefix + ".weight"], a=0, mode="fan_in")
        sd[prefix + ".weight"] *= 0.1
        sd[prefix + ".bias"].zero_()

    conv("head.feature_extraction", nf, 3)
    reinit("head.feature_extraction")
    for i, nb in enumerate(blocks):
        for j in range(nb):
            p = "body_%d.res_blocks.%d.body" % (i, j)
            conv(p + ".0", nf, nf)
            conv(p + ".2", nf, nf)
            reinit(p + ".0")
            reinit(p + ".2")
        p = "body_%d.leg.recon_block" % i
        conv(p + ".0", nf, nf)
        conv(p + ".2", 48, nf)
        reinit(p + ".0")
        reinit(p + ".2")
    if v2:
        conv("tail.merge_conv", nf, nf * len(blocks))
        conv("tail.recon_block.0", nf, nf)
        conv("tail.recon_block.2", 48, nf)
        reinit("tail.recon_block.0")
        reinit("tail.recon_block.2")
        reinit("tail.merge_conv")
    return sd


def train_steps(sd, x, truth, blocks, steps=1, lr=4e-4, v2=False):
    """`steps` iterations of models/LarvaNet.py:98-114 (loss, backward, AdamW step).  Returns the
    loss of every step and the gradients of the LAST step; sd is updated in place."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=lr)
    losses, grads = [], None
    for _ in range(steps):
        loss = multi_exit_loss(params, x, truth, blocks, v2=v2)
        opt.zero_grad()
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in params.items()}
        opt.step()
        losses.append(float(loss.item()))
    for k in sd:
        sd[k] = params[k].detach()
    return losses, grads


def image_psnr_protocol(out, truth):
    """validate.py:17-27 on one image pair (CHW float arrays): round half to even, clip to 0..255, uint8; the truth
    cropped top-left to the output's size; 10 log10(255^2 / mean squared error) over all RGB values."""
    import numpy as np
    o8 = np.clip(np.round(out), 0, 255).astype(np.uint8)
    t8 = np.clip(np.round(truth), 0, 255).astype(np.uint8)[:, :o8.shape[1], :o8.shape[2]]
    d = np.float32(t8) - np.float32(o8)
    return 10.0 * np.log10(255.0 ** 2 / np.mean(np.power(d, 2)))   # (a float32 scalar, as in the reference)


def train_trajectory(sd, batches, steps, blocks, val_pairs, volume_per_step, val_volume, lr=4e-4, lr_decay=0.5,
                     patience=3, cooldown=6, threshold=1e-3, min_lr=1e-8, v2=False):
    """The plugin's step loop with its bookkeeping (models/LarvaNet.py:98-139): global_step / temp_volume counters,
    the multi-exit step, validation at step 1 (:116-117) and whenever temp_volume >= val_volume (:119-124:
    total_volume += temp_volume, temp_volume = 0, validate_for_train, save), validate_for_train's mean PSNR over
    the validation pairs stepping ReduceLROnPlateau(mode max, abs threshold; :90-92,141-161), and the file name
    save() writes (:183-185).  Step s trains on batches[s % len(batches)].  Returns a dict of per-step losses,
    learning rates, volumes, and per-validation steps / PSNRs / checkpoint names; sd is updated in place."""
    import numpy as np
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=lr)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=lr_decay, patience=patience,
                                                       cooldown=cooldown, threshold=threshold, threshold_mode="abs",
                                                       min_lr=min_lr)
    rec = {"losses": [], "lrs": [], "total_volume": [], "temp_volume": [], "val_steps": [], "psnrs": [], "ckpt_names": []}
    global_step, total_volume, temp_volume = 0, 0.0, 0

    def validate():
        vals = []
        with torch.no_grad():
            for lr_img, hr_img in val_pairs:
                x = torch.from_numpy(np.asarray(lr_img, np.float32))[None].to(next(iter(params.values())).dtype)
                out = (forward_v2(params, x, blocks) if v2 else forward(params, x, blocks))[0].numpy()
                vals.append(image_psnr_protocol(out, hr_img))
        avg = np.mean(vals)   # (models/LarvaNet.py:158: the mean of float32 scalars stays float32)
        rec["val_steps"].append(global_step)
        rec["psnrs"].append(float(avg))
        sched.step(avg)

    for s in range(steps):
        global_step += 1
        temp_volume += volume_per_step
        x, truth = batches[s % len(batches)]
        loss = multi_exit_loss(params, x, truth, blocks, v2=v2)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if global_step == 1:
            validate()
        if temp_volume >= val_volume:
            total_volume += temp_volume
            temp_volume = 0
            validate()
            rec["ckpt_names"].append("model_step%d_vol%.0fG.pth" % (global_step, total_volume / 1e9))
        rec["losses"].append(float(loss.item()))
        rec["lrs"].append(opt.param_groups[0]["lr"])
        rec["total_volume"].append(total_volume)
        rec["temp_volume"].append(temp_volume)
    for k in sd:
        sd[k] = params[k].detach()
    rec["scheduler"] = sched
    return rec
