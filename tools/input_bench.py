#!/usr/bin/env python3
"""Throughput of the GPU input step on synthetic decoded images: the kernel alone and the whole main-process path
(pinned copy + H2D + kernel).  Prints one JSON line.
    python tools/input_bench.py [--n 128] [--src 512] [--size 224]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from util.gpu_input import FMOW_RGB_MEAN, FMOW_RGB_STD, GpuAugment, pack_uint8, sample_transform_params

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128); ap.add_argument("--src", type=int, default=512); ap.add_argument("--size", type=int, default=224)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
imgs = [torch.randint(0, 256, (a.src, a.src, 3), generator=g, dtype=torch.uint8) for _ in range(a.n)]
torch.manual_seed(0)
params = [sample_transform_params(a.src, a.src) for _ in range(a.n)]
aug = GpuAugment(a.size)
packed = pack_uint8(imgs)  # what collate_uint8 hands over from the loader workers
for _ in range(3):
    out = aug(packed, params)
torch.cuda.synchronize()
# (1) main-process cost of a loader iteration: copy into pinned memory + H2D + kernel
t0 = time.perf_counter()
pending = aug.stage(packed, params)          # PrefetchLoader's pattern: batch k+1 is staged before batch k is consumed
for _ in range(a.iters):
    nxt = aug.stage(packed, params)
    out = aug.finish(pending)
    pending = nxt
out = aug.finish(pending)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / (a.iters + 1)
# (2) kernel alone, source resident in HBM
from csmae_hip import ops
src = packed.data.cuda(); meta = torch.tensor(params, dtype=torch.int32).cuda(); dst = torch.empty_like(out)
ops.augment_u8(src, meta, aug.mean, aug.inv_std, dst); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.augment_u8(src, meta, aug.mean, aug.inv_std, dst)
e1.record(); torch.cuda.synchronize()
t_k = e0.elapsed_time(e1) / a.iters * 1e-3
read_b = sum(p[4] * p[5] * 3 for p in params); write_b = a.n * 3 * a.size * a.size * 4
print(json.dumps({"metric": "input step images/s (decoded uint8 -> normalised fp32 crops)", "n": a.n, "src": a.src, "size": a.size,
                  "kernel_images_per_s": round(a.n / t_k, 1), "kernel_us": round(t_k * 1e6, 1),
                  "kernel_hbm_gbps_algorithmic": round((read_b + write_b) / t_k / 1e9, 1),
                  "pinned_copy_h2d_kernel_images_per_s": round(a.n / t_all, 1), "h2d_mb_per_batch": round(packed.data.numel() / 2 ** 20, 1),
                  }))
