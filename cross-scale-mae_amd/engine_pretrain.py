"""One epoch of Cross-Scale MAE pre-training — same signature, schedule, meters and return value as the reference's
`engine_pretrain.train_one_epoch` (engine_pretrain.py:18-101), re-timed for the MI355X:

* the reference synchronises host and device four times per iteration (loss.item(), GradScaler found-inf, cuda.synchronize(),
  all_reduce_mean().item()); here the per-iteration losses stay on the device and are drained every `print_freq` iterations,
  so kernels of step k+1 are enqueued while step k still runs.  Meter contents and the non-finite-loss ValueError are the same,
  the error is just raised at the next drain instead of the same iteration — on every rank at the same drain — and the fused
  optimizer skips, on the device, every update whose loss was not finite (csmae_adamw's gate), so weights and AdamW moments
  are still the last good ones when the error surfaces;
* `torch.cuda.synchronize()` is not required (the loop is device agnostic and is unit-tested on the CPU with a stub model);
* with accum_iter > 1 the gradient all-reduce happens only on the update micro-step (DDP `no_sync`), not on every one.
"""
import contextlib
import math
from typing import Iterable

import torch

import util.lr_sched as lr_sched
import util.misc as misc

PRINT_FREQ = 20


def _autocast(device):
    """`with torch.cuda.amp.autocast()` of the reference: on the MI355X path it selects the bf16 MFMA kernels."""
    dev = torch.device(device)
    if dev.type == "cuda":
        return torch.autocast("cuda", dtype=torch.bfloat16)
    return contextlib.nullcontext()


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device, epoch: int,
                    loss_scaler, log_writer=None, args=None):
    model.train(True)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = f"Epoch: [{epoch}]"
    accum_iter = args.accum_iter
    optimizer.zero_grad()
    if log_writer is not None:
        print(f"log_dir: {log_writer.log_dir}")
    n_iters = len(data_loader)
    pending = []  # (data_iter_step, device loss, lr)
    world = misc.get_world_size()

    def drain():
        if not pending:
            return
        values = torch.stack([p[1].detach().float().reshape(()) for p in pending])
        if world > 1:
            reduced = values.clone()
            torch.distributed.all_reduce(reduced)
            reduced = (reduced / world).tolist()
        else:
            reduced = None
        values = values.tolist()
        for k, (it, _, lr) in enumerate(pending):
            loss_value = values[k]
            # (world > 1: the rank-mean decides, so that every rank raises at the same drain instead of leaving its peers in a collective)
            if not math.isfinite(loss_value) or (reduced is not None and not math.isfinite(reduced[k])):
                print(f"Loss is {loss_value}, stopping training")
                raise ValueError(f"Loss is {loss_value}, stopping training")
            metric_logger.update(loss=loss_value)
            metric_logger.update(lr=lr)
            if log_writer is not None and (it + 1) % accum_iter == 0:
                epoch_1000x = int((it / n_iters + epoch) * 1000)
                log_writer.add_scalar("train_loss", reduced[k] if reduced else loss_value, epoch_1000x)
                log_writer.add_scalar("lr", lr, epoch_1000x)
        pending.clear()

    for data_iter_step, (samples, _) in enumerate(metric_logger.log_every(data_loader, PRINT_FREQ, header)):
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / n_iters + epoch, args)
        samples = samples.to(device, non_blocking=True)
        update = (data_iter_step + 1) % accum_iter == 0
        sync_ctx = contextlib.nullcontext() if update or not hasattr(model, "no_sync") else model.no_sync()
        with sync_ctx:
            with _autocast(device):
                loss, _, _ = model(samples, mask_ratio=args.mask_ratio)
            pending.append((data_iter_step, loss, optimizer.param_groups[0]["lr"]))
            loss = loss / accum_iter
            loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=update)
        if update:
            optimizer.zero_grad()
        if data_iter_step % PRINT_FREQ == 0 or data_iter_step == n_iters - 1:
            drain()  # exactly the iterations on which log_every prints the meters
    drain()
    if hasattr(optimizer, "join"):
        optimizer.join()   # (FusedAdamW(overlap=True): whatever reads the weights after this epoch — checkpointing, evaluation — is behind the last step)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
