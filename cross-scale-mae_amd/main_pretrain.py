#!/usr/bin/env python3
"""Cross-Scale MAE pre-training driver for the MI355X path — the reference's `main_pretrain.py` command line (all 40 flags with
identical names, types and defaults: main_pretrain.py:41-356), model factory call (`models_mae.__dict__[args.model](**vars(args))`,
:398), LR scaling (:406-412), AdamW param groups (:426-427), checkpoint cadence (:579-587) and `log.jsonl` (:631-634).

Differences, all additive:
  * `--dataset_type synthetic` (+ `--synthetic_len`, `--input_channels`) feeds in-memory N(0,1) batches generated on the device —
    the fMoW / EuroSAT / NAIP / COCO loaders of `util/datasets.py` need rasterio/fiona/torchvision and are outside the hot-path scope
    (SURVEY.md §2 row 16); selecting them raises with that explanation;
  * the model is wrapped in `csmae_hip.parallel.DataParallel` (flat-buffer RCCL all-reduce overlapped with backward) instead of DDP,
    and the optimizer is the fused HIP AdamW with torch.optim.AdamW's state layout;
  * W&B / TensorBoard and the matplotlib reconstruction plots at checkpoint epochs (:589-626) are not wired (observability / plotting
    UI, not on the measured path — SURVEY.md §2 rows 13, 19);
  * `--honor_start_epoch` resumes the epoch counter (the reference ignores --start_epoch: main_pretrain.py:554-555).

    torchrun --nproc_per_node=8 main_pretrain.py --model mae_vit_base_MsLdCeCd --dataset_type synthetic --batch_size 128 --epochs 1
"""
import argparse
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

import models_mae
import util.misc as misc
from engine_pretrain import train_one_epoch
from util.misc import NativeScalerWithGradNormCount as NativeScaler


def nullable_string(val):
    return val if val else None


def get_args_parser():
    p = argparse.ArgumentParser("Cross-MAE pre-training", add_help=False)
    p.add_argument("--batch_size", type=int, default=512, help="Batch size per GPU (effective batch size is batch_size * accum_iter * # gpus")
    p.add_argument("--epochs", default=200, type=int)
    p.add_argument("--accum_iter", type=int, default=1)
    p.add_argument("--model", default="mae_vit_base", type=str, metavar="MODEL")
    p.add_argument("--input_size", type=int, default=224)
    p.add_argument("--patch_size", type=str, default=16)
    p.add_argument("--print_level", type=int, default=1)
    p.add_argument("--mask_ratio", type=float, default=0.75)
    p.add_argument("--attn_name", type=str, default="scaled_dot_product",
                   choices=["scaled_dot_product", "shunted", "linformer", "orthoformer", "nystrom", "fourier_mix", "local"])
    p.add_argument("--use_xformers", action="store_true")
    p.set_defaults(use_xformers=False)
    p.add_argument("--ffn_name", type=str, default="MLP", choices=["MLP", "FusedMLP"])
    p.add_argument("--spatial_mask", action="store_true", default=False)
    p.add_argument("--loss", type=str, default="mse", choices=["mse", "mae", "l1", "l2", "bce", "ssim", "ms_ssim", "mse_ssim", "mse_ms_ssim"])
    p.add_argument("--norm_pix_loss", action="store_true")
    p.set_defaults(norm_pix_loss=False)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--lr", type=float, default=None, metavar="LR")
    p.add_argument("--blr", type=float, default=5e-5, metavar="LR")
    p.add_argument("--min_lr", type=float, default=0.0, metavar="LR")
    p.add_argument("--warmup_epochs", type=int, default=40, metavar="N")
    p.add_argument("--train_path", default="./train.csv", type=str)
    p.add_argument("--dataset_type", type=str, default="fmow_rgb", choices=["fmow_rgb", "euro_sat", "naip", "coco", "synthetic"])
    p.add_argument("--masked_bands", type=int, nargs="+", default=None)
    p.add_argument("--dropped_bands", type=int, nargs="+", default=None)
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--output_dir_base", type=str, default="./out")
    p.add_argument("--val_img_path", type=str, default="./images/")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", type=nullable_string, default=None)
    p.add_argument("--start_epoch", type=int, default=0, metavar="N")
    p.add_argument("--wandb_entity", type=str, default="utk-iccv23")
    p.add_argument("--wandb_project", type=nullable_string, default=None)
    p.add_argument("--wandb_id", type=nullable_string, default=None)
    p.add_argument("--num_workers", type=int, default=os.cpu_count())
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=os.getenv("LOCAL_RANK", 0), type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    # ---- additive flags of the MI355X build
    p.add_argument("--synthetic_len", type=int, default=64, help="iterations per epoch of the synthetic loader")
    p.add_argument("--input_channels", type=int, default=None, help="bands of the synthetic loader / model (reference constructors default to 3)")
    p.add_argument("--honor_start_epoch", action="store_true", help="start the epoch loop at --start_epoch / the resumed epoch")
    p.add_argument("--grad_comm_dtype", type=str, default="fp32", choices=["auto", "fp32", "bf16"],
                   help="RCCL gradient payload: fp32 (default: what the reference's DDP exchanges), bf16 (half the xGMI bytes; opt-in), auto = bf16 when the engine computes in bf16 / fp8")
    return p


class SyntheticLoader:
    """In-memory repeat loader: yields (samples[N,C,S,S] float32 ~ N(0,1) on `device`, None) — the loader contract of
    engine_pretrain.py:41 without any host->device traffic (SURVEY.md §8d measurement protocol)."""

    def __init__(self, batch, channels, size, length, device, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        self.samples = torch.randn(batch, channels, size, size, device=device, generator=g)
        self.length = length
        self.sampler = self

    def set_epoch(self, epoch):
        pass

    def __len__(self):
        return self.length

    def __iter__(self):
        for _ in range(self.length):
            yield self.samples, None


def output_dir_name(args, model_name=None):
    """Naming template of main_pretrain.py:450-468."""
    name = "_".join([args.model, f"xformers-{args.attn_name}-{args.ffn_name}" if args.use_xformers else f"{args.attn_name}",
                     f"i{args.input_size}-p{args.patch_size}-mr{args.mask_ratio}", f"e{args.epochs}-we{args.warmup_epochs}",
                     f"b{args.batch_size}-a{args.accum_iter}", f"{args.loss}{'-normpix' if args.norm_pix_loss else ''}", f"lr{args.lr}",
                     args.dataset_type])
    return name


def protect_output_dir(output_dir, model_name, resume, distributed):
    """Existing runs are never overwritten (main_pretrain.py:470-490).  Without --resume: a single-process run moves on to
    `out_<name>+1`, `+2`, ... while the directory exists; a distributed run keeps its name (every rank must agree on it) and refuses
    to start when checkpoints are already in there."""
    import glob
    if resume is not None:
        return output_dir
    if not distributed:
        while os.path.exists(output_dir):
            if not glob.glob(os.path.join(output_dir, "*.pth")):
                print(f"INFO: {output_dir} already exists, but contains no .pth files. You may want to delete it.")
            number = os.path.basename(output_dir).split("+")[-1]
            number = int(number) + 1 if number.isdigit() else 1
            output_dir = os.path.join(os.path.dirname(output_dir), f"out_{model_name}+{number}")
    elif glob.glob(os.path.join(output_dir, "*.pth")):
        raise ValueError(f"ERROR: {output_dir} already exists and contains .pth files. Checkpoints would be overwritten.")
    return output_dir


def main(args):
    misc.init_distributed_mode(args)
    print(f"job dir: {os.path.dirname(os.path.realpath(__file__))}")
    print(f"{args}".replace(", ", ",\n"))
    device = torch.device(args.device)
    seed = args.seed + misc.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)

    if args.dataset_type not in ("synthetic", "fmow_rgb"):
        raise NotImplementedError(f"--dataset_type {args.dataset_type}: the reference's multi-band readers (util/datasets.py) depend on rasterio / fiona "
                                  "and are outside the MI355X hot-path scope; use --dataset_type synthetic / fmow_rgb or pass your own "
                                  "iterable of (samples, _) to engine_pretrain.train_one_epoch")
    channels = args.input_channels or 3
    if device.type == "cuda":
        torch.cuda.set_device(getattr(args, "gpu", 0) if args.distributed else torch.cuda.current_device())
        device = torch.device("cuda", torch.cuda.current_device())
    if args.dataset_type == "synthetic":
        data_loader_train = SyntheticLoader(args.batch_size, channels, args.input_size, args.synthetic_len, device, seed)
    else:
        # fMoW-RGB CSV (util/datasets.py:161-206): the workers only decode; flips, normalisation and the bicubic RandomResizedCrop of
        # the reference's transform (util/datasets.py:120-136) run in one HIP kernel behind pinned double-buffered uint8 staging
        from util.gpu_input import CsvImageDataset, GpuAugment, PrefetchLoader, collate_uint8
        dataset = CsvImageDataset(args.train_path)
        if args.distributed:
            sampler = torch.utils.data.DistributedSampler(dataset, num_replicas=misc.get_world_size(), rank=misc.get_rank(), shuffle=True)
        else:
            sampler = torch.utils.data.RandomSampler(dataset)
        raw = torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                          pin_memory=False, drop_last=True, collate_fn=collate_uint8)
        data_loader_train = PrefetchLoader(raw, GpuAugment(args.input_size, device=device))
        data_loader_train.sampler = sampler

    kwargs = dict(vars(args))
    if args.input_channels is None:
        kwargs.pop("input_channels")
    model = models_mae.__dict__[args.model](**kwargs)
    model.to(device)
    model_without_ddp = model
    print(f"Model = {model_without_ddp}")

    batch_size_eff = args.batch_size * args.accum_iter * misc.get_world_size()
    print("accumulate grad iterations: %d" % args.accum_iter)
    print("effective batch size: %d" % batch_size_eff)
    if args.lr is None:
        args.lr = args.blr * batch_size_eff / 256
    print("base lr: %.2e" % (args.lr * 256 / batch_size_eff))
    print("actual lr: %.2e" % args.lr)

    from csmae_hip.optim import FusedAdamW, add_weight_decay
    from csmae_hip.parallel import DataParallel
    if args.distributed:
        model = DataParallel(model, device_ids=[args.gpu], find_unused_parameters=True,
                             comm_dtype={"auto": "auto", "bf16": torch.bfloat16, "fp32": None}[args.grad_comm_dtype])
        model_without_ddp = model.module
    # overlap: the step is enqueued on its own stream and the next forward pass orders each layer behind the launch that steps its weights (optim.py)
    optimizer = FusedAdamW(add_weight_decay(model_without_ddp, args.weight_decay), lr=args.lr, betas=(0.9, 0.95), overlap=True)
    print(optimizer)
    loss_scaler = NativeScaler()
    misc.load_model(args=args, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler)
    print(f"Trainable parameters: {sum(p.numel() for p in model.parameters() if p.requires_grad)}")

    model_name = output_dir_name(args)
    if args.output_dir is None:
        args.output_dir = f"out_{model_name}"
    if args.output_dir_base is not None:
        args.output_dir = os.path.join(args.output_dir_base, args.output_dir)
    args.output_dir = protect_output_dir(args.output_dir, model_name, resume=args.resume, distributed=args.distributed)
    print(f"Output directory: {args.output_dir}")
    if misc.is_main_process():
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)

    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    first = args.start_epoch if args.honor_start_epoch else 0
    for epoch in range(first, args.epochs):
        if args.distributed:
            data_loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, data_loader_train, optimizer, device, epoch, loss_scaler, log_writer=None, args=args)
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch}
        print(f"Train stats: {log_stats}")
        if args.output_dir and (epoch % 25 == 0 or epoch + 1 == args.epochs):
            misc.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler, epoch=epoch)
        if args.output_dir and misc.is_main_process():
            with open(os.path.join(args.output_dir, "log.jsonl"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print(f"Training time {datetime.timedelta(seconds=int(time.time() - start_time))}")


if __name__ == "__main__":
    main(get_args_parser().parse_args())
