#!/usr/bin/env python3
"""Timing of the ssim-family loss head at the headline geometry: two views of N=128, 3x224x224, p=16.
    python tools/ssim_bench.py [--iters 20]
Prints per-call times of csmae_ssim_fwd / csmae_ssim_bwd for ssim (1 level) and ms_ssim (5 levels) and the HBM rate against the
algorithmic bytes (every plane read / written once per kernel that needs it)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import ops  # noqa: E402


def level_sizes(S, levels):
    h, out = S, []
    for _ in range(levels):
        out.append(h)
        h = (h + 2 * (h & 1) - 2) // 2 + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--size", type=int, default=224)
    a = ap.parse_args()
    N, C, S, p = a.n, 3, a.size, 16
    B2, L, P = 2 * N, (S // p) ** 2, p * p * C
    dev = "cuda"
    img0, img1 = torch.randn(N, C, S, S, device=dev), torch.randn(N, C, S, S, device=dev)
    pred = torch.randn(B2 * (L + 1), P, device=dev)
    mask = (torch.rand(B2, L, device=dev) > 0.25).float()
    gout = torch.ones(1, device=dev)
    terms = torch.empty(2, device=dev)
    extra = torch.empty(B2 * L, P, device=dev)
    for levels in (1, 5):
        ws = torch.empty(ops.ssim_workspace_floats(B2, C, S, p, levels), device=dev)
        hs = level_sizes(S, levels)
        planes = B2 * C
        px = [planes * h * h * 4 for h in hs]
        img_b, pred_b = B2 * C * S * S * 4, B2 * L * P * 4
        # fwd: min/max sweeps read pred + image; prepare reads both again and writes X0, Y0; each level reads X, Y; pooling reads X, Y, writes the next X, Y
        fwd_b = 2 * (img_b + pred_b) + 2 * px[0] + sum(2 * b for b in px) + sum(2 * px[i] + 2 * px[i + 1] for i in range(levels - 1))
        # bwd: each level reads X, Y (+ the next level's gradient) and writes its gradient; the pred sweep reads D0 + pred, writes extra; tie sweep reads pred
        bwd_b = sum(3 * b for b in px) + sum(px[1:]) + px[0] + 3 * pred_b
        for name, fn, nbytes in (("fwd", lambda: ops.ssim_fwd(levels, False, img0, img1, pred, mask, ws, terms, B2, N, C, S, p), fwd_b),
                                 ("bwd", lambda: ops.ssim_bwd(levels, pred, mask, gout, 1.0, ws, extra, B2, N, C, S, p), bwd_b)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{'ssim' if levels == 1 else 'ms_ssim':8s} {name}  B2={B2} {C}x{S}x{S}  {ms * 1e3:9.1f} us   {nbytes / 1e6:8.1f} MB algorithmic  {nbytes / ms / 1e9:6.2f} TB/s   "
                  f"workspace {ws.numel() * 4 / 2**20:.0f} MiB")
        print("terms", terms.tolist())


if __name__ == "__main__":
    main()
