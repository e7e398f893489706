"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.

BASELINE.json configs[0]: the reference's own CPU-runnable case, `EDSR-baseline x4, batch 4, 48x48 LR patches on
PyTorch-CPU via train.py` (models/edsr.py, train.py:83-105).  EDSR is not on the hot path ("NOT a
kernel target") and gets no HIP kernels; this file restates its training step with the torch CPU operators the
reference calls, functional style over a state_dict with the reference's key names, so that (a) the plumbing the
config names is pinned by a fixture generated from the imported reference (tests/golden F15) and (b) bench.py can
time it on the GPU box's host cores as the CPU-baseline leg BASELINE.md section 3 promises.
"""
import math

import torch
import torch.nn.functional as F


def init_state_dict(features=64, res_blocks=16, scale=4, seed=None):
    """EDSRModule's parameters in construction order with nn.Conv2d's default initialisation
    (models/edsr.py:177-192).  MeanShift (models/edsr.py:129-137) sets `weight_data` / `bias_data`, not the
    convolution's own weight / bias, so both mean-shift layers are frozen 1x1 convs with nn.Conv2d's RANDOM default
    weights -- restated as is."""
    if seed is not None:
        torch.manual_seed(seed)
    sd = {}

    def conv(prefix, cout, cin, k):
        w = torch.empty(cout, cin, k, k)
        torch.nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        b = torch.empty(cout)
        bound = 1.0 / (cin * k * k) ** 0.5
        torch.nn.init.uniform_(b, -bound, bound)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = w, b

    conv("mean_shift", 3, 3, 1)
    conv("first_conv", features, 3, 3)
    for i in range(res_blocks):
        conv("res_blocks.%d.body.0" % i, features, features, 3)
        conv("res_blocks.%d.body.2" % i, features, features, 3)
    conv("after_res_conv", features, features, 3)
    if scale in (2, 4, 8):
        for s in range(int(math.log(scale, 2))):
            conv("upsample.body.%d" % (2 * s), 4 * features, features, 3)
    else:
        conv("upsample.body.0", 9 * features, features, 3)
    conv("final_conv", 3, features, 3)
    conv("mean_inverse_shift", 3, 3, 1)
    return sd


FROZEN = ("mean_shift.weight", "mean_shift.bias", "mean_inverse_shift.weight", "mean_inverse_shift.bias")


def forward(sd, x, res_blocks, scale=4, res_weight=1.0):
    """EDSRModule.forward (models/edsr.py:194-207)."""
    x = F.conv2d(x, sd["mean_shift.weight"], sd["mean_shift.bias"])
    x = F.conv2d(x, sd["first_conv.weight"], sd["first_conv.bias"], padding=1)
    res = x
    for i in range(res_blocks):
        p = "res_blocks.%d.body" % i
        h = F.relu(F.conv2d(res, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1))
        res = torch.add(res, F.conv2d(h, sd[p + ".2.weight"], sd[p + ".2.bias"], padding=1).mul(res_weight))
    res = F.conv2d(res, sd["after_res_conv.weight"], sd["after_res_conv.bias"], padding=1)
    x = torch.add(x, res)
    if scale in (2, 4, 8):
        for s in range(int(math.log(scale, 2))):
            p = "upsample.body.%d" % (2 * s)
            x = F.pixel_shuffle(F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1), 2)
    else:
        x = F.pixel_shuffle(F.conv2d(x, sd["upsample.body.0.weight"], sd["upsample.body.0.bias"], padding=1), 3)
    x = F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"], padding=1)
    return F.conv2d(x, sd["mean_inverse_shift.weight"], sd["mean_inverse_shift.bias"])


def make_trainer(sd, res_blocks, scale=4, lr=1e-4, lr_decay=0.5, lr_decay_steps=200000, res_weight=1.0, global_step=0):
    """EDSR.prepare(is_training=True) + EDSR.train_step (models/edsr.py:36-58, 75-108): optim.Adam over the trainable
    parameters, L1 loss, the step-decayed learning rate set before every step.  Returns step(x, truth) -> loss."""
    params = {k: (v.clone().requires_grad_(True) if k not in FROZEN else v.clone()) for k, v in sd.items()}
    state = {"global_step": global_step}

    def learning_rate():
        return lr * (lr_decay ** (state["global_step"] // lr_decay_steps))

    opt = torch.optim.Adam([v for k, v in params.items() if k not in FROZEN], lr=learning_rate())

    def step(x, truth):
        loss = F.l1_loss(forward(params, x, res_blocks, scale, res_weight), truth)
        for group in opt.param_groups:
            group["lr"] = learning_rate()
        opt.zero_grad()
        loss.backward()
        opt.step()
        state["global_step"] += 1
        return float(loss.item())

    step.params = params
    step.state = state
    return step
