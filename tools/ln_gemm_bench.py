#!/usr/bin/env python3
"""The full-row GEMM + LayerNorm kernels (csrc/gemm_ln.hip) against the kernel pairs they replace, alone on the chip, on the decoder's shapes
(ViT-B/16, N = 128, two views; per view for the forward products): HIP-event timing per launch.
    python tools/ln_gemm_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import EPI_RESID, ops  # noqa: E402

MD = 50432


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
    N = 512
    g, b, bias = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1, torch.randn(N, device=dev) * 0.1
    for name, M, K in (("dec.proj + norm2 (view)", MD // 2, 512), ("dec.fc2 + norm1 (view)", MD // 2, 2048), ("dec.proj + norm2", MD, 512), ("dec.fc2 + norm1", MD, 2048)):
        A = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
        Wk = W.view(N, K // 32, 32).permute(1, 0, 2).contiguous().reshape(-1)
        resid = torch.randn(M, N, device=dev).to(bf)
        x, y = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        t_f = timeit(lambda: ops.gemm_ln_fwd(A, Wk, bias, resid, x, g, b, y, mean, rstd), a.iters)
        t_g = timeit(lambda: ops.gemm_ks(A, Wk, W, x, bias=bias, epilogue=EPI_RESID, resid=resid), a.iters)
        t_l = timeit(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd), a.iters)
        fl = 2.0 * M * N * K
        print(f"{name:26s} M={M:6d} K={K:5d} | fused {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF/s) | gemm {t_g:7.1f} + ln_fwd {t_l:6.1f} = {t_g + t_l:7.1f} us | {t_f / (t_g + t_l):.2f}x")
    for name, M, K in (("dec.fc1-dX + norm2'", MD, 2048), ("dec.qkv-dX + norm1'", MD, 1536)):
        dY = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(K, N, device=dev) * K ** -0.5).to(bf)
        xs, dres = torch.randn(M, N, device=dev).to(bf), torch.randn(M, N, device=dev).to(bf)
        y, t1, dx = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        ops.layernorm_fwd(xs, g, b, y, mean, rstd)
        part = torch.empty(1024 * 2 * N, device=dev)
        t_f = timeit(lambda: ops.gemm_ln_bwd(dY, W, xs, mean, rstd, g, dres, dx, partial_ws=part), a.iters)
        t_g = timeit(lambda: ops.gemm(dY, W, t1, trans_b=True), a.iters)
        t_l = timeit(lambda: ops.layernorm_bwd(t1, xs, mean, rstd, g, dx, None, None, dres_in=dres, partial_ws=part), a.iters)
        fl = 2.0 * M * N * K
        print(f"{name:26s} M={M:6d} K={K:5d} | fused {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF/s) | gemm {t_g:7.1f} + ln_bwd {t_l:6.1f} = {t_g + t_l:7.1f} us | {t_f / (t_g + t_l):.2f}x")


if __name__ == "__main__":
    main()
