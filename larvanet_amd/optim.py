"""AdamW for the flat-buffer layout: a torch.optim.AdamW (so `.param_groups[0]['lr']`,
ReduceLROnPlateau, state_dict() keep working exactly as with the reference's optimizer,
models/LarvaNet.py:86-92) whose step() is ONE HIP launch over the flat parameter / gradient /
moment buffers instead of a multi-tensor pass over 82 tensors.  If the parameters or gradients
are not (or no longer) contiguous views of the flat buffers it falls back to torch's own step()."""
import torch

from . import kernels as K


def flatten_parameters(module):
    """Re-seat every trainable parameter as a view of one flat fp32 buffer (same values, same
    nn.Parameter objects, same state_dict keys). Returns the flat buffer."""
    params = [p for p in module.parameters() if p.requires_grad]
    flat = torch.empty(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
    off = 0
    with torch.no_grad():
        for p in params:
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += p.numel()
    return flat


class FlatAdamW(torch.optim.AdamW):
    def __init__(self, params, flat_params, grad_bucket, lr=1e-3, **kw):
        params = list(params)
        super().__init__(params, lr=lr, **kw)
        self._flat_p = flat_params
        self._bucket = grad_bucket
        self._m = torch.zeros_like(flat_params)
        self._v = torch.zeros_like(flat_params)
        self._t = 0
        self._plist = params
        # (src, dst) 0-d tensors: the next flat step() also copies src into dst (the step's loss), then forgets it
        self.copy_scalar = None
        self.mean_scale = 1.0   # (not "grad_scale": torch.optim reserves that attribute for AMP) gradients are multiplied by this inside the step (1/world_size: mean over ranks)

    def _flat_ok(self):
        if self._flat_p is None or self._bucket is None or not self._bucket.intact():
            return False
        off = 0
        base = self._flat_p.data_ptr()
        for p in self._plist:
            if p.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or len(self.param_groups) != 1 or not self._flat_ok():
            return self._fallback_step(closure)
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            return self._fallback_step(closure)
        self._t += 1
        b1, b2 = g["betas"]
        copy, self.copy_scalar = self.copy_scalar, None
        K.adamw_step_host(self._flat_p, self._bucket.flat, self._m, self._v, self._t, float(g["lr"]), b1, b2,
                          g["eps"], g["weight_decay"], self.mean_scale, copy=copy)
        return None

    def training_state(self):
        """Everything needed to continue training bit-for-bit (the reference saves weights only)."""
        if self._flat_p is not None:
            return {"flat": True, "t": self._t, "m": self._m.detach().cpu().clone(), "v": self._v.detach().cpu().clone(),
                    "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        return {"flat": False, "torch": super().state_dict()}

    def load_training_state(self, st):
        if st.get("flat") and self._flat_p is not None:
            self._t = int(st["t"])
            self._m.copy_(st["m"].to(self._m.device))
            self._v.copy_(st["v"].to(self._v.device))
            for g, saved in zip(self.param_groups, st["param_groups"]):
                g.update(saved)
        elif not st.get("flat"):
            super().load_state_dict(st["torch"])
        else:
            raise RuntimeError("larvanet_amd: flat optimizer state cannot be loaded into a per-tensor optimizer")

    def _fallback_step(self, closure):
        if self.mean_scale != 1.0:   # consumed once: the caller sets it again if it still applies
            for p in self._plist:
                if p.grad is not None:
                    p.grad.mul_(self.mean_scale)
            self.mean_scale = 1.0
        # hand the moments over to torch's per-tensor state once, then stay on torch's path
        if self._t > 0 and not self.state:
            off = 0
            for p in self._plist:
                n = p.numel()
                self.state[p] = {"step": torch.tensor(float(self._t)),
                                 "exp_avg": self._m[off:off + n].view_as(p).clone(),
                                 "exp_avg_sq": self._v[off:off + n].view_as(p).clone()}
                off += n
        self._flat_p = None
        return super().step(closure)
