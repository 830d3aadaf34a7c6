"""Fused multi-tensor AdamW over the model's flat fp32 parameter / gradient buffers.

Behaviourally `torch.optim.AdamW(param_groups, lr, betas=(0.9, 0.95))` as wired by the reference (main_pretrain.py:426-427,
timm `add_weight_decay` grouping); `state_dict()` keeps torch's per-parameter layout (`step`, `exp_avg`, `exp_avg_sq`) so
checkpoints stay interchangeable (util/misc.py:364-370).  One HIP launch per parameter group reads p, g, m, v once (28 B/param)."""
from __future__ import annotations

import contextlib
import math
import os

import torch

from . import ops, trace


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
    CHUNKS = int(os.environ.get("CSMAE_OPT_CHUNKS", "16"))   # overlap mode: launches per parameter group, one event each ...
    CHUNK_MIN_TILES = 256  # ... of at least a million elements (the bias / norm group stays one launch)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, overlap=False):
        """`overlap=True`: the step is enqueued on the optimizer's own stream, in `CHUNKS` launches per group in parameter order, and returns with the
        CURRENT stream NOT ordered behind it: the next forward pass of the model's engine orders each layer behind the chunk that steps its weights
        (Engine._opt_gate), so the optimizer's HBM-bound 0.7 ms run under the stem and the first encoder blocks instead of alone on the chip.  Whoever
        reads parameters or optimizer state on another stream before that forward pass calls `join()` first (state_dict() does; zero_grad(set_to_none=
        False) does).  Off by default: `step()` then is ordered on the current stream like torch.optim's."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None
        self._plans = {}
        self._m = self._v = None
        self.overlap = bool(overlap)
        self._ostream = None
        self._oevents = []

    def _bind(self):
        from .engine import FlatParams
        first = self.param_groups[0]["params"][0] if self.param_groups[0]["params"] else self.param_groups[1]["params"][0]
        flat = FlatParams.owner_of(first)
        if flat is None:
            raise RuntimeError("FusedAdamW: parameters are not homed in a csmae_hip flat buffer yet — run one forward pass of the model on the "
                               "GPU before the first optimizer step")
        if flat is not self._flat:
            self._sync_steps()  # (the step counts live in the plans that are about to be dropped)
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
        # fp8 mode (the last forward ran an fp8 engine): the block Linear weights are stepped through csmae_adamw_fp8 in 64 x 64 sub-blocks, which writes
        # their fp8 mirrors (W8 / W8^T) in the same pass — no re-quantisation launches at the start of the next forward
        f8 = bool(getattr(flat, "fp8_active", False)) and getattr(flat, "w8", None) is not None
        key = (gi, active, f8)
        if key not in self._plans:
            self._sync_steps()
            for old in [k for k in self._plans if k[0] == gi]:  # the group's active set changed: its old plan (and counter) is superseded
                del self._plans[old]
            offs, cnts, params, ks, t8, f8_names, b8 = [], [], [], [], [], set(), []
            ks_names = getattr(flat, "ks_names", None) if getattr(flat, "w_ks", None) is not None else None
            for p in group["params"]:
                if p.grad is None:
                    continue
                params.append(p)
                off, n, shape = flat.slot_of(p)
                name = flat._by_id[id(p)]
                if f8 and name in flat.w8_idx and len(shape) == 2 and shape[0] % 64 == 0 and shape[1] % 64 == 0:
                    wi = flat.w8_idx[name]
                    b8.append((len(t8), name))
                    t8.extend((off, shape[0], shape[1], n0, k0, wi) for n0 in range(0, shape[0], 64) for k0 in range(0, shape[1], 64))
                    f8_names.add(name)
                    continue
                # the K-slab mirror of a block Linear weight (Engine._refresh_ks) is written by the same launch: (weight offset, N, K) per tile
                row = (off, shape[0], shape[1]) if (ks_names is not None and flat._by_id[id(p)] in ks_names) else (0, 0, 0)
                for o in range(0, n, self.TILE):
                    offs.append(off + o)
                    cnts.append(min(self.TILE, n - o))
                    ks.append(row)
            dev = flat.p.device
            wd = float(group["weight_decay"])
            has_ks = any(r[2] for r in ks)
            self._plans[key] = dict(off=torch.tensor(offs, dtype=torch.long, device=dev), cnt=torch.tensor(cnts, dtype=torch.int32, device=dev),
                                    wd=torch.full((len(offs),), wd, device=dev), wd_host=wd, params=params,
                                    t8=torch.tensor(t8, dtype=torch.long, device=dev).reshape(-1, 6) if t8 else None, f8_names=f8_names,
                                    ks=torch.tensor(ks, dtype=torch.long, device=dev).reshape(-1, 3) if has_ks else None,
                                    ks_names={flat._by_id[id(q)] for q in params} & set(ks_names) if has_ks else set(),
                                    step=self._common_step(params), chunks=self._chunks(flat, params, offs, f8_names),
                                    chunks8=self._cut(b8, len(t8), self.CHUNKS, 1))
        return self._plans[key]

    def _chunks(self, flat, params, offs, skip):
        """Overlap mode: the plan's tiles cut into up to CHUNKS contiguous launches on parameter boundaries -> [(tile_lo, tile_hi, names)]."""
        bounds, t = [], 0   # (first tile, name) per parameter, in plan order
        for p in params:
            name = flat._by_id[id(p)]
            if name in skip:
                continue
            bounds.append((t, name))
            t += -(-flat.slot_of(p)[1] // self.TILE)
        assert t == len(offs)
        return self._cut(bounds, t, self.CHUNKS, self.CHUNK_MIN_TILES)

    @staticmethod
    def _cut(bounds, t, chunks, min_tiles):
        nch = min(chunks, max(1, t // min_tiles))
        per, out = max(1, -(-t // nch)), []
        for t0, name in bounds:
            if not out or (t0 - out[-1][0] >= per and len(out) < nch):
                out.append([t0, t, [name]])
                if len(out) > 1:
                    out[-2][1] = t0
            else:
                out[-1][2].append(name)
        return [tuple(c) for c in out]

    def _common_step(self, params):
        """torch.optim.AdamW counts steps per parameter; one launch per group needs one bias correction.  Parameters of a group that
        have been stepped a different number of times (possible only when the set of parameters with gradients changed mid-run)
        cannot share a launch: say so instead of silently using the first one's count."""
        steps = {int(self.state[p]["step"]) for p in params}
        if len(steps) > 1:
            raise RuntimeError(f"FusedAdamW: parameters of one group have different step counts {sorted(steps)} (the set of parameters "
                               "with gradients changed during training); use torch.optim.AdamW for such a schedule")
        return steps.pop() if steps else 0

    @torch.no_grad()
    def step(self, closure=None):
        """No host<->device synchronisation and no staging copies in here: the hyper-parameters are kernel arguments, the step counter
        and the weight decay live on the host, so the CPU keeps enqueueing the next step while the GPU is still in this one's backward.
        The launches are gated on the device: when the loss behind these gradients was not finite (the engine leaves it in
        `flat.gate`, all-reduced with the gradients) the kernels return without touching weights, moments or the bf16 mirror — the
        reference raises before `backward` (engine_pretrain.py:56-58), this loop only notices at its next loss drain."""
        loss = closure() if closure is not None else None
        trace.push("csmae.optimizer")
        try:
            self._step()
        finally:
            trace.pop()
        return loss

    def _step(self):
        flat = self._bind()
        g0 = flat.g.data_ptr()
        ks_before = getattr(flat, "ks_stamp", None) is not None and flat.ks_stamp == (flat.lp_stamp, flat.raw_writes)   # K-slab mirror consistent on entry
        ks_written, f8_written = set(), set()
        f8_before = getattr(flat, "w8_stamp", None) is not None and flat.w8_stamp == (flat.version_stamp(), flat.raw_writes)   # fp8 mirrors consistent on entry
        todo = []
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group, flat)
            params = plan["params"]
            if not params:
                continue
            for p in params:  # a gradient tensor that is not the engine's view of the flat buffer: bring it home
                if p.grad.data_ptr() - g0 != flat.slot_of(p)[0] * 4:
                    off, n, _ = flat.slot_of(p)
                    flat.g[off:off + n].view(p.shape).copy_(p.grad)
            wd = float(group["weight_decay"])
            if wd != plan["wd_host"]:
                plan["wd"].fill_(wd)
                plan["wd_host"] = wd
            todo.append((group, plan, wd))
        # overlap mode (see __init__): the launches go to the optimizer's stream, behind everything enqueued on the current one (the backward pass,
        # the gradient exchange's join); one event per launch for the engine's next forward pass to order its layers behind
        ov = self.overlap and flat.p.is_cuda and ops._timer is None and not os.environ.get("CSMAE_OPT_MAIN")
        self.join()                   # (a previous overlapped step nobody ordered behind: this one's launches must be; no-op otherwise)
        st, pending = None, None
        if not ov:
            flat.opt_pending = None   # (this step is ordered on the current stream: nothing for a forward pass to gate on)
        if ov:
            if self._ostream is None:
                self._ostream = torch.cuda.Stream()
            self._ostream.wait_stream(torch.cuda.current_stream())
            st = self._ostream.cuda_stream
            pending = dict(events=[], chunk_of={}, waited={}, stream=self._ostream)
        for group, plan, wd in todo:
            plan["step"] += 1
            step = plan["step"]
            b1, b2 = group["betas"]
            ks_ok = plan["ks"] is not None and flat.w_lp is not None and getattr(flat, "w_ks", None) is not None
            if plan["off"].numel():
                for lo, hi, names in (plan["chunks"] if ov else [(0, plan["off"].numel(), ())]):
                    ops.adamw(plan["off"][lo:hi], plan["cnt"][lo:hi], plan["wd"][lo:hi], flat.p, flat.g, self._m, self._v, group["lr"], b1, b2, group["eps"], step,
                              p_lp=flat.w_lp, gate=flat.gate, tile_ks=plan["ks"][lo:hi] if ks_ok else None, p_ks=flat.w_ks if ks_ok else None, st=st)
                    if ov:
                        self._mark(pending, names)
            if plan["t8"] is not None:
                with torch.cuda.stream(self._ostream) if ov else contextlib.nullcontext():
                    flat.w8_amax[1].zero_()
                for lo, hi, names in (plan["chunks8"] if ov else [(0, plan["t8"].shape[0], ())]):
                    ops.adamw_fp8(plan["t8"][lo:hi], wd, flat.p, flat.g, self._m, self._v, group["lr"], b1, b2, group["eps"], step, flat.w_lp, flat.gate, flat.w8, flat.w8t,
                                  flat.w8_amax[0], flat.w8_amax[1], flat.w8_dq, st=st)
                    if ov:
                        self._mark(pending, names)
                f8_written |= plan["f8_names"]
            # the kernel wrote the masters (and the bf16 mirror) behind torch's version counters: mirrors derived from it (fp8) are stale.  The K-slab
            # mirror is not, when this launch wrote it for every weight that has one and it was consistent before (Engine._refresh_ks's stamp)
            flat.raw_writes += 1
            if ks_ok:
                ks_written |= plan["ks_names"]
            self._dirty_steps = True
        if ov and pending["events"]:
            flat.opt_pending = pending
        if ks_before and ks_written and ks_written == set(flat.ks_names):   # every K-slab mirror was re-written by the launches above: still consistent
            flat.ks_stamp = (flat.lp_stamp, flat.raw_writes)
        if f8_written:
            flat.w8_amax.reverse()   # what this step measured is what the next rewrite scales with
            if f8_before and f8_written == set(flat.w8_idx):   # every fp8 mirror was re-written from the stepped masters: consistent, Engine._refresh_fp8 has nothing to do
                flat.w8_stamp = (flat.version_stamp(), flat.raw_writes)
            else:
                flat.w8_stamp = None

    def _mark(self, pending, names):
        """An event behind the launch just enqueued on the optimizer's stream; `names` = the parameters it stepped."""
        k = len(pending["events"])
        while len(self._oevents) <= k:
            self._oevents.append(torch.cuda.Event())
        self._oevents[k].record(self._ostream)
        pending["events"].append(self._oevents[k])
        pending["chunk_of"].update((n, k) for n in names)

    def join(self):
        """Order the current stream behind an overlapped step still in flight (no host synchronisation).  No-op otherwise."""
        flat = self._flat
        pend = getattr(flat, "opt_pending", None) if flat is not None else None
        if pend is not None:
            cur = torch.cuda.current_stream()
            last = len(pend["events"]) - 1
            if pend["waited"].get(cur.cuda_stream, -1) < last:
                cur.wait_event(pend["events"][last])
                pend["waited"][cur.cuda_stream] = last

    def zero_grad(self, set_to_none: bool = True):
        if not set_to_none:
            self.join()   # (the zero-fill runs on the current stream: behind the step's reads of the gradients)
        return super().zero_grad(set_to_none=set_to_none)

    def _sync_steps(self):
        """torch's per-parameter `step` entries are refreshed lazily (state_dict / checkpointing), not 250 tensors per step."""
        if getattr(self, "_dirty_steps", False):
            for plan in self._plans.values():
                for p in plan["params"]:
                    self.state[p]["step"] = torch.tensor(float(plan["step"]))
            self._dirty_steps = False

    def state_dict(self):
        self.join()
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None  # re-alias the loaded moments into the flat buffers at the next step
        self._dirty_steps = False
