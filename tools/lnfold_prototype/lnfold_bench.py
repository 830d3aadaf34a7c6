#!/usr/bin/env python3
"""Epilogue cost of the LayerNorm fold on the step's forward shapes (ViT-B/16, N = 128, per forward stream = one view): plain vs folded
consumer (qkv, fc1 + GELU), plain vs statistics-emitting residual producer (proj, fc2), alone on the chip.  HIP-event timing per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import EPI_GELU, EPI_RESID, ops  # noqa: E402

ME, MD = 6400, 25216   # rows of one view


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    bf = dict(device="cuda", dtype=torch.bfloat16)
    for stack, M, D in (("enc", ME, 768), ("dec", MD, 512)):
        parts = (D + 255) // 256
        x = (torch.randn(M, D, device="cuda") * 1.5).to(torch.bfloat16)
        st = torch.empty(parts, M, 2, device="cuda")
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        for name, N, gelu in (("qkv", 3 * D, False), ("fc1+gelu", 4 * D, True)):
            w = (torch.randn(N, D, device="cuda") * D ** -0.5).to(torch.bfloat16)
            b, c = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
            out = torch.empty(M, N, **bf)
            aux = torch.empty(M, N, device="cuda", dtype=torch.uint8) if gelu else None
            # valid statistics for the fold
            xf = x.float()
            for p in range(parts):
                seg = xf[:, p * 256:(p + 1) * 256]
                st[p, :, 0], st[p, :, 1] = seg.sum(1), (seg * seg).sum(1)
            epi = EPI_GELU if gelu else 0
            t0 = timeit(lambda: ops.gemm(x, w, out, bias=b, epilogue=epi, aux=aux))
            t1 = timeit(lambda: ops.gemm_lnfold(x, w, out, c, b, st, parts, mean, rstd, epilogue=epi, aux=aux))
            fl = 2.0 * M * N * D
            print(f"{stack}.{name:9s} {M}x{N}x{D}: plain {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF)  folded {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF)  {t1 - t0:+6.1f} us")
        for name, K in (("proj", D), ("fc2", 4 * D)):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(D, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(D, device="cuda")
            out = torch.empty(M, D, **bf)
            t0 = timeit(lambda: ops.gemm(a, w, out, bias=b, epilogue=EPI_RESID, resid=x))
            t1 = timeit(lambda: ops.gemm_resid_stats(a, w, out, b, x, st))
            fl = 2.0 * M * D * K
            print(f"{stack}.{name:9s} {M}x{D}x{K}: plain {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF)  +stats {t1:7.1f} us ({fl / t1 / 1e6:6.0f} TF)  {t1 - t0:+6.1f} us")
        y = torch.empty(M, D, **bf)
        g, be = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
        t = timeit(lambda: ops.layernorm_fwd(x, g, be, y, mean, rstd))
        print(f"{stack}.ln_fwd    {M}x{D}: {t:7.1f} us")


if __name__ == "__main__":
    main()
