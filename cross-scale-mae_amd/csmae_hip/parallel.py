"""Batch-sharded data parallelism for one 8-GPU MI355X node: one process per GPU, RCCL over xGMI.

Replaces `torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)` of the reference
(main_pretrain.py:417-421).  The path has exactly one bulk exchange per step: the gradient mean.  Because all
gradients live in one flat fp32 buffer in parameter-registration order, the exchange is a handful of large contiguous
all-reduces issued on a side stream while the encoder's backward GEMMs are still running (decoder + heads are ready first and
are 24 % of the bytes).  xGMI is point-to-point (7 links x ~153 GB/s), so few large messages beat DDP's 25 MB buckets.

`GradSync` is device-agnostic (works on CPU tensors with gloo) so the N>1 logic is covered by world-size-2 tests without a GPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import trace


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


class GradSync:
    """Mean-all-reduce of contiguous ranges of a flat gradient buffer, optionally on a side stream."""

    def __init__(self, flat_grad: torch.Tensor, group=None, comm_dtype: Optional[torch.dtype] = None, force: bool = False):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if is_dist() else 1
        self.force = force and is_dist()   # issue the collectives even at world size 1 (stream / event ordering under test on one GPU)
        self.comm_dtype = comm_dtype
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self._staging = torch.empty(flat_grad.numel(), dtype=comm_dtype, device=flat_grad.device) if comm_dtype not in (None, flat_grad.dtype) else None
        self._pending = False
        self.timing = False   # bench.py: HIP events around every bucket's exchange on the exchange stream (bucket_times())
        self._timed = []
        self.issued = []   # (lo, hi) of the ranges exchanged by the current / last backward pass: cleared by the first reduce_range() after a finish() (tests read it)
        self._fresh = True
        # gloo has no AVG; RCCL does
        self._avg = self.cuda and is_dist() and dist.get_backend(group) == "nccl"

    def reduce_range(self, lo: int, hi: int, also=None):
        """Enqueue mean-all-reduce of g[lo:hi].  On GPU it runs on the side stream after everything already enqueued on the
        current stream and on `also` (the engine's weight-gradient stream): the kernels that produced g[lo:hi].  The producers' streams
        are not made to wait for each other — only the exchange waits."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        if self._fresh:   # (one (lo, hi) per bucket and step: without this the list grows for the length of the training run)
            self.issued.clear()
            self._fresh = False
        self.issued.append((lo, hi))
        seg = self.g[lo:hi]
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            if also is not None:
                self.stream.wait_stream(also)
            with torch.cuda.stream(self.stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                self._reduce(seg, lo, hi)
                if self.timing:
                    e1.record(self.stream)
                    self._timed.append((lo, hi, e0, e1))
            self._pending = True
        else:
            self._reduce(seg, lo, hi)

    def _reduce(self, seg, lo, hi):
        if self._staging is not None:  # bf16 payload (opt-in): halves xGMI bytes (SURVEY.md §8e budget)
            st = self._staging[lo:hi]
            hip = self.cuda and seg.dtype == torch.float32 and st.dtype == torch.bfloat16   # the two casts are HIP kernels on the exchange stream
            if hip:
                from . import ops
                ops.cast_bf16(seg, st)
            else:   # (CPU / gloo tests of the control flow)
                st.copy_(seg)
            dist.all_reduce(st, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group)
            if hip:
                ops.cast_f32(st, seg)
            else:
                seg.copy_(st)
            if not self._avg:
                seg.div_(self.world)
        elif self._avg:
            dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            seg.div_(self.world)

    def bucket_times(self):
        """[(elements, ms)] per exchanged range since timing was switched on, in issue order: the time between the exchange stream reaching
        the bucket (its producers done) and the collective's completion on that stream — the ring time plus whatever the collective's
        kernels wait for CUs.  Synchronises."""
        if self.cuda:
            torch.cuda.synchronize()
        out = [(hi - lo, e0.elapsed_time(e1)) for lo, hi, e0, e1 in self._timed]
        self._timed = []
        return out

    def finish(self):
        """Make the reduced gradients visible to the current stream (the optimizer's stream)."""
        if self._pending:
            torch.cuda.current_stream().wait_stream(self.stream)
            self._pending = False
        self._fresh = True

    def broadcast(self, tensors: List[torch.Tensor], src: int = 0):
        if self.world == 1 and not self.force:
            return
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)


def bucket_ranges(slots: dict, names: List[str], n_encoder: int, enc_per_bucket: int = 4, taper: bool = False,
                  tail_extra: int = 0) -> List[Tuple[str, int, int]]:
    """Contiguous flat ranges in gradient-READY order (reverse registration order): tail (decoder + heads), encoder layers in
    groups from the last to the first, then the stem (tokens, patch embed, decoder_embed).  With `taper` the groups shrink
    towards the end of the backward pass (…, 4, 2, 1, 1): whatever is reduced after the last backward kernel is exposed time, so
    the final messages should be small (one ViT-B layer = 28 MB fp32 ~ 0.3 ms on the ring) while the early ones stay large."""
    def start(prefix):
        for n in names:
            if n.startswith(prefix):
                return slots[n][0]
        return None
    total = max(o + ((n + 7) // 8) * 8 for o, n, _ in slots.values())
    out = []
    dec0 = start("decoder.0.")
    enc0 = start("encoder.0.")
    out.append(("tail", dec0, total + tail_extra))  # (+ the update-gate slot behind the last parameter, FlatParams.gate)
    hi = dec0
    i = n_encoder
    while i > 0:
        size = enc_per_bucket
        if taper:
            size = 1 if i <= 2 else (2 if i <= 4 else enc_per_bucket)
        j = max(0, i - size)
        lo = start(f"encoder.{j}.")
        out.append((("enc", j), lo, hi))  # final once encoder layer j's backward has been enqueued
        hi, i = lo, j
    out.append(("stem", 0, enc0))
    return out


class FlatBuffers:
    """The module's buffers (BatchNorm running statistics and step counter of the predictor: 2 x L floats + one int64) re-homed
    into ONE byte buffer, so that DDP's per-forward `broadcast_buffers` is a single small broadcast instead of one per tensor
    (three latency-bound RCCL launches on the critical path in front of every forward)."""

    def __init__(self, module: torch.nn.Module):
        self.items = [(n, b) for n, b in module.named_buffers() if b is not None]
        self.raw = None
        self.views = []
        if not self.items:
            return
        dev = self.items[0][1].device
        offs, off = [], 0
        for _, b in self.items:
            offs.append(off)
            off += (b.numel() * b.element_size() + 15) // 16 * 16
        self.raw = torch.zeros(off, dtype=torch.uint8, device=dev)
        with torch.no_grad():
            for (_, b), o in zip(self.items, offs):
                v = self.raw[o:o + b.numel() * b.element_size()].view(b.dtype).view(b.shape)
                v.copy_(b)
                b.data = v   # in-place updates by the kernels (running statistics) and by load_state_dict now land in the flat buffer
                self.views.append(v)

    def still_homed(self) -> bool:
        return all(b.data_ptr() == v.data_ptr() and b.device == v.device for (_, b), v in zip(self.items, self.views))


class DataParallel(torch.nn.Module):
    """Wrapper with DDP's surface (`.module`, forward passthrough, `no_sync()`): hooks the engine's backward so gradient
    ranges are all-reduced as soon as they are final."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, comm_dtype=None, force_collectives: bool = False, **_):
        """comm_dtype: gradient payload on the wire.  None (default) = fp32, what the reference's DDP exchanges (main_pretrain.py:417-421)
        and the only payload until an 8-GPU run has validated another one; torch.bfloat16 halves the xGMI bytes (opt-in); "auto" = bf16
        when the engine that owns the flat buffers computes in bf16 / fp8 (it has a bf16 weight mirror), fp32 in parity mode."""
        super().__init__()
        self.module = module
        self.comm_dtype = comm_dtype
        self.force_collectives = force_collectives
        self.require_backward_grad_sync = True
        self._sync: Optional[GradSync] = None
        module.__dict__["_dp"] = self  # plain attribute (not a registered submodule: that would make the module tree cyclic)

    def _ensure(self, flat):
        if self._sync is None or self._sync.g is not flat.g:
            cd = self.comm_dtype
            if isinstance(cd, str):   # "auto": from the engine's own numerics mode (a bf16 weight mirror exists), never from ambient autocast state
                cd = torch.bfloat16 if flat.w_lp is not None else None
            self._sync = GradSync(flat.g, comm_dtype=cd, force=self.force_collectives)
            # rank-0 parameters and buffers win (DDP constructor semantics, main_pretrain.py:418-420)
            self._sync.broadcast([flat.p])
            flat.mark_changed()   # the broadcast wrote the masters behind the Parameters' version counters: the bf16 mirror must be recast
            self._broadcast_buffers()
            n_enc = len(self.module.encoder)
            self._ranges = {name: (lo, hi) for name, lo, hi in bucket_ranges(flat.slots, flat.names, n_enc, taper=True,
                                                                               tail_extra=flat.g.numel() - flat.total)}
        return self._sync

    def _broadcast_buffers(self):
        fb = self.__dict__.get("_fbuf")
        if fb is None or not fb.still_homed():   # first use, or the module was moved (`.to()` re-allocates every buffer)
            fb = self.__dict__["_fbuf"] = FlatBuffers(self.module)
        if fb.raw is not None:
            self._sync.broadcast([fb.raw])

    # called by Engine.backward
    def grads_ready(self, flat, name, also=None):
        if not self.require_backward_grad_sync or not is_dist():
            return
        sync = self._ensure(flat)
        if name in self._ranges:
            with trace.range_("csmae.exchange"):
                sync.reduce_range(*self._ranges[name], also=also)

    def wants(self, name) -> bool:
        """True if `name` closes a bucket (lets the engine skip joining its weight-gradient stream otherwise)."""
        if self._sync is None and is_dist() and self.module._flat is not None:
            self._ensure(self.module._flat)
        return self.require_backward_grad_sync and is_dist() and self._sync is not None and name in self._ranges

    def backward_done(self, flat):
        if self._sync is not None:
            self._sync.finish()

    def forward(self, *args, **kwargs):
        if is_dist():
            if self.module._flat is None or not self.module._flat.still_homed():
                self.module._engine(args[0])  # homes the parameters in the flat buffers so they can be broadcast before first use
            self._ensure(self.module._flat)
            if self.module.training:  # DDP broadcast_buffers=True: BatchNorm statistics follow rank 0
                self._broadcast_buffers()
        return self.module(*args, **kwargs)

    class _NoSync:
        def __init__(self, dp):
            self.dp = dp

        def __enter__(self):
            self.old = self.dp.require_backward_grad_sync
            self.dp.require_backward_grad_sync = False

        def __exit__(self, *a):
            self.dp.require_backward_grad_sync = self.old

    def no_sync(self):
        return DataParallel._NoSync(self)
