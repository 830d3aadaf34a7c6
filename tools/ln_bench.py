#!/usr/bin/env python3
"""LayerNorm forward / backward kernels alone on the chip, on the block shapes of the presets (bf16 residual stream, deferred parameter gradients), with and
without the fused fp8 copy: HIP-event time per launch and the algorithmic bytes over it.  CSMAE_DEBUG=ln_bwd_unpacked selects the generic backward instance.
    python tools/ln_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev, bf = "cuda", torch.bfloat16
    for name, M, D in (("ViT-B enc", 12800, 768), ("ViT-B dec", 50432, 512), ("ViT-L enc", 12800, 1024), ("ViT-H/14 enc (256)", 33280, 1280), ("ViT-H/14 dec (256)", 131584, 512)):
        x, dy, dres = (torch.randn(M, D, device=dev).to(bf) for _ in range(3))
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g, b = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev) * 0.1
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        part = torch.empty(1024 * 2 * D, device=dev)
        q = torch.empty(M, D, device=dev, dtype=torch.uint8)
        amax = [torch.full((64,), 4.0, device=dev), torch.zeros(64, device=dev)]
        dq = torch.zeros(1, device=dev)
        unit = M * D * 2 / 1e6   # MB per bf16 tensor
        for emit in (None, (q, ops.FP8_E4M3, amax[0], amax[1], dq)):
            tf = timeit(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, emit=emit), a.iters)
            eb = (q, ops.FP8_E5M2, amax[0], amax[1], dq) if emit else None
            tb = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, None, None, dres_in=dres, partial_ws=part, emit=eb), a.iters)
            bf_, bb = unit * (2 + (0.5 if emit else 0)), unit * (4 + (0.5 if emit else 0))
            print(f"{name:20s} M={M:6d} D={D:4d} fp8 copy {'yes' if emit else 'no ':3s} | fwd {tf:6.1f} us {bf_ / tf:5.2f} TB/s | bwd {tb:6.1f} us {bb / tb:5.2f} TB/s")


if __name__ == "__main__":
    main()
