"""Host-side helpers of the pre-training loop (reference util/misc.py): meters, distributed init, the loss-scaler callable,
checkpoint save/load with the reference's dict layout.  No device math lives here except the optional gradient-norm read."""
import builtins
import datetime
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist


class SmoothedValue:
    """Window median/avg + global average (util/misc.py:26-86)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter
        self.MB = 1024.0 * 1024.0

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        """Yields the items; prints a progress line every `print_freq` iterations; records time_epoch / time_step at the end
        (util/misc.py:125-190)."""
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable)
        width = len(str(n))
        gpu = torch.cuda.is_available()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                parts = [header, f"[{i:{width}d}/{n}]", f"eta: {eta}", str(self), f"iter_time: {iter_time}", f"data_time: {data_time}"]
                if gpu:
                    mem = torch.cuda.max_memory_allocated() / self.MB
                    parts.append(f"memory: {mem:.0f}")
                    self.update(memory_alloc=mem)
                print(self.delimiter.join(parts))
            end = time.time()
        total = time.time() - start
        self.update(time_epoch=total, time_step=total / max(n, 1))
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


def setup_for_distributed(is_master):
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False) or get_world_size() > 8
        if is_master or force:
            builtin_print(f"[{datetime.datetime.now().time()}] ", end="")
            builtin_print(*args, **kwargs)

    builtins.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """env:// rendezvous from torchrun / OMPI / SLURM variables (util/misc.py:255-296); backend nccl == RCCL on ROCm."""
    if getattr(args, "dist_on_itp", False):
        args.rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
        os.environ.update(LOCAL_RANK=str(args.gpu), RANK=str(args.rank), WORLD_SIZE=str(args.world_size))
    elif "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size, args.gpu = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % torch.cuda.device_count()
    else:
        print("Not using distributed mode")
        setup_for_distributed(is_master=True)
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available() and str(getattr(args, "device", "cuda")).startswith("cuda")
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    # (RCCL's kernels run beside GEMM workgroups that each own a whole CU: a high-priority queue lets a collective take the next CU that frees up)
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    print(f"| distributed init (rank {args.rank}): {args.dist_url}, gpu {args.gpu}, world size {args.world_size}", flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.0)
    if norm_type == float("inf"):
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack([torch.norm(g, norm_type) for g in grads]), norm_type)


def clip_grad_norm_(parameters, max_norm):
    """`torch.nn.utils.clip_grad_norm_(parameters, max_norm)` (util/misc.py:314-316; `max_norm=None`: `get_grad_norm_`, :318) -> the total
    2-norm as a 0-dim tensor, gradients scaled in place by min(1, max_norm / (norm + 1e-6)).  When the gradients are the engine's
    views of the flat gradient buffer — the case on the MI355X path — this is two streaming HIP kernels over that one buffer with the
    coefficient read from device memory (no host sync, no ~250-tensor foreach), also when the user has frozen some parameters (their
    slots are cleared by the engine); anything else (a foreign model, a subset of the model's parameters, CPU tests of the loop) goes
    through torch."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    flat = None
    if params and params[0].is_cuda:
        from csmae_hip.engine import FlatParams
        flat = FlatParams.owner_of(params[0])
        if flat is not None:
            g0, views = flat.g.data_ptr(), 0
            for p in params:
                name = flat._by_id.get(id(p))
                if name is None or p.grad.data_ptr() != g0 + flat.slots[name][0] * 4:
                    flat = None
                    break
                views += 1
            # every parameter of the model that has a gradient must be in the call (the buffer is normed as a whole)
            if flat is not None and views != sum(1 for q in flat.params.values() if q.grad is not None):
                flat = None
            # (the buffer is normed as a whole: slots of parameters without a gradient hold zeros — never written (sin-cos tables, the
            # discarded encoder_norm) or cleared at the end of Engine.backward (parameters the user froze))
    if flat is None:
        if max_norm is None:
            return get_grad_norm_(params)
        return torch.nn.utils.clip_grad_norm_(params, max_norm)
    from csmae_hip import ops
    scratch = flat.__dict__.get("_norm_scratch")
    if scratch is None:
        scratch = flat.__dict__["_norm_scratch"] = torch.empty(1024 + 2, device=flat.g.device, dtype=torch.float32)
    out = scratch[1024:]
    ops.clip_grad_norm(flat.g[: flat.total], -1.0 if max_norm is None else float(max_norm), scratch, out)
    return out[0].clone()


class NativeScalerWithGradNormCount:
    """Same callable / state_dict contract as the reference (util/misc.py:299-335).  The MI355X path computes in bf16 (or fp32),
    which needs no loss scaling, so the wrapped GradScaler is disabled: backward -> optional clip -> optimizer.step().
    The reference computes the global gradient norm on every update and the engine ignores it; here it is only computed when
    clipping is requested or `compute_grad_norm=True` (it costs one extra read of every gradient) — by HIP kernels over the flat
    gradient buffer (`clip_grad_norm_` above).
    fp16 + loss scaling (what the reference's `torch.cuda.amp.autocast()` + GradScaler runs, engine_pretrain.py:52, util/misc.py:303)
    is deliberately not emulated: the throughput path is bf16 (fp32 exponent range: no overflow to scale away), the parity path is
    fp32; an fp16 mode would add a third numerics mode that matches neither the fp32 oracle nor the bf16 MFMA path."""
    state_dict_key = "amp_scaler"

    def __init__(self, compute_grad_norm=False):
        self._scaler = torch.amp.GradScaler("cuda", enabled=False)
        self.compute_grad_norm = compute_grad_norm

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        norm = None
        if update_grad:
            if clip_grad is not None:
                assert parameters is not None
                norm = clip_grad_norm_(parameters, clip_grad)
            elif self.compute_grad_norm:
                norm = clip_grad_norm_(parameters, None)
            optimizer.step()
        return norm

    def state_dict(self):
        return self._scaler.state_dict()

    def load_state_dict(self, state_dict):
        self._scaler.load_state_dict(state_dict)


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler):
    """<output_dir>/checkpoint-<epoch>.pth with the reference's dict layout (util/misc.py:358-372)."""
    path = Path(args.output_dir) / f"checkpoint-{epoch}.pth"
    save_on_master({"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
                    "scaler": loss_scaler.state_dict() if loss_scaler is not None else {}, "args": args}, path)


def load_model(args, model_without_ddp, optimizer, loss_scaler, device=None):
    if not getattr(args, "resume", None):
        print("Not resuming from checkpoint")
        return
    if args.resume.startswith("https"):
        checkpoint = torch.hub.load_state_dict_from_url(args.resume, map_location="cpu", check_hash=True)
    else:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
    model_without_ddp.load_state_dict(checkpoint["model"], strict=False)
    if device is not None:
        model_without_ddp.to(device)
    print(f"Resuming from checkpoint: {args.resume}")
    if "optimizer" in checkpoint and "epoch" in checkpoint and not getattr(args, "eval", False):
        optimizer.load_state_dict(checkpoint["optimizer"])
        args.start_epoch = checkpoint["epoch"] + 1
        if "scaler" in checkpoint and loss_scaler is not None:
            loss_scaler.load_state_dict(checkpoint["scaler"])
        print("With optim & sched!")


def all_reduce_mean(x):
    world_size = get_world_size()
    if world_size <= 1:
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(x, device=dev)
    dist.all_reduce(t)
    return (t / world_size).item()
