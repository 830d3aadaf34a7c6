"""Fused multi-tensor AdamW over the model's flat fp32 parameter / gradient buffers.

Behaviourally `torch.optim.AdamW(param_groups, lr, betas=(0.9, 0.95))` as wired by the reference (main_pretrain.py:426-427,
timm `add_weight_decay` grouping); `state_dict()` keeps torch's per-parameter layout (`step`, `exp_avg`, `exp_avg_sq`) so
checkpoints stay interchangeable (util/misc.py:364-370).  One HIP launch per parameter group reads p, g, m, v once (28 B/param)."""
from __future__ import annotations

import math

import torch

from . import ops


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """timm.optim.optim_factory.add_weight_decay as used at main_pretrain.py:426: 1-D tensors and biases are not decayed."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if p.ndim == 1 or name.endswith(".bias") or name in skip_list else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FusedAdamW(torch.optim.Optimizer):
    TILE = 4096

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None
        self._plans = {}
        self._m = self._v = None

    def _bind(self):
        from .engine import FlatParams
        first = self.param_groups[0]["params"][0] if self.param_groups[0]["params"] else self.param_groups[1]["params"][0]
        flat = FlatParams.owner_of(first)
        if flat is None:
            raise RuntimeError("FusedAdamW: parameters are not homed in a csmae_hip flat buffer yet — run one forward pass of the model on the "
                               "GPU before the first optimizer step")
        if flat is not self._flat:
            self._flat, self._plans = flat, {}
            self._m = torch.zeros_like(flat.p)
            self._v = torch.zeros_like(flat.p)
            for group in self.param_groups:
                for p in group["params"]:
                    old = self.state.get(p, {})
                    off, n, shape = flat.slot_of(p)
                    st = {"step": old.get("step", torch.tensor(0.0)), "exp_avg": self._m[off:off + n].view(shape),
                          "exp_avg_sq": self._v[off:off + n].view(shape)}
                    if "exp_avg" in old:
                        st["exp_avg"].copy_(old["exp_avg"])
                        st["exp_avg_sq"].copy_(old["exp_avg_sq"])
                    self.state[p] = st
        return flat

    def _plan(self, gi, group, flat):
        active = tuple(id(p) for p in group["params"] if p.grad is not None)
        key = (gi, active)
        if key not in self._plans:
            offs, cnts = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                off, n, _ = flat.slot_of(p)
                if p.grad.data_ptr() != flat.g.data_ptr() + off * 4:
                    flat.g[off:off + n].view(p.shape).copy_(p.grad)  # foreign gradient tensor: bring it home
                for o in range(0, n, self.TILE):
                    offs.append(off + o)
                    cnts.append(min(self.TILE, n - o))
            dev = flat.p.device
            self._plans[key] = (torch.tensor(offs, dtype=torch.long, device=dev), torch.tensor(cnts, dtype=torch.int32, device=dev),
                                torch.full((len(offs),), float(group["weight_decay"]), device=dev), torch.zeros(6, device=dev),
                                torch.zeros(6).pin_memory() if torch.cuda.is_available() else torch.zeros(6))
        return self._plans[key]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        flat = self._bind()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            step = int(self.state[params[0]]["step"]) + 1
            toff, tcnt, twd, hyper, host = self._plan(gi, group, flat)
            if twd.numel() and float(group["weight_decay"]) != float(twd[0]):
                twd.fill_(float(group["weight_decay"]))
            b1, b2 = group["betas"]
            host.copy_(torch.tensor([group["lr"], b1, b2, group["eps"], 1 - b1 ** step, 1 - b2 ** step]))
            hyper.copy_(host, non_blocking=True)
            ops.adamw(toff, tcnt, twd, flat.p, flat.g, self._m, self._v, hyper, p_lp=flat.w_lp)
            for p in params:
                self.state[p]["step"] = torch.tensor(float(step))
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None  # re-alias the loaded moments into the flat buffers at the next step
