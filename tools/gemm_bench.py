#!/usr/bin/env python3
"""Micro-benchmark of the GEMM family on the step's real shapes (ViT-B/16, N=128, two views): HIP-event timing per launch.
    python tools/gemm_bench.py [--iters 20] [--only dec]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import ops  # noqa: E402

ME, MD = 12800, 50432
CASES = [
    # name, layout, M, N, K, epilogue, out_dtype
    ("enc.qkv      fwd", "NT", ME, 2304, 768, 0, torch.bfloat16),
    ("enc.proj     fwd", "NT", ME, 768, 768, 2, torch.bfloat16),
    ("enc.fc1+gelu fwd", "NT", ME, 3072, 768, 1, torch.bfloat16),
    ("enc.fc2      fwd", "NT", ME, 768, 3072, 2, torch.bfloat16),
    ("dec.qkv      fwd", "NT", MD, 1536, 512, 0, torch.bfloat16),
    ("dec.proj     fwd", "NT", MD, 512, 512, 2, torch.bfloat16),
    ("dec.fc1+gelu fwd", "NT", MD, 2048, 512, 1, torch.bfloat16),
    ("dec.fc2      fwd", "NT", MD, 512, 2048, 2, torch.bfloat16),
    ("enc.fc2  dX+dgelu", "NN", ME, 3072, 768, 3, torch.bfloat16),
    ("enc.fc1  dX", "NN", ME, 768, 3072, 0, torch.bfloat16),
    ("enc.qkv  dX", "NN", ME, 768, 2304, 0, torch.bfloat16),
    ("dec.fc2  dX+dgelu", "NN", MD, 2048, 512, 3, torch.bfloat16),
    ("dec.fc1  dX", "NN", MD, 512, 2048, 0, torch.bfloat16),
    ("dec.qkv  dX", "NN", MD, 512, 1536, 0, torch.bfloat16),
    ("enc.proj dX", "NN", ME, 768, 768, 0, torch.bfloat16),
    ("dec.proj dX", "NN", MD, 512, 512, 0, torch.bfloat16),
    ("enc.fc1  dW", "TN", 3072, 768, ME, 4, torch.float32),
    ("enc.qkv  dW", "TN", 2304, 768, ME, 4, torch.float32),
    ("enc.proj dW", "TN", 768, 768, ME, 4, torch.float32),
    ("enc.fc2  dW", "TN", 768, 3072, ME, 4, torch.float32),
    ("dec.qkv  dW", "TN", 1536, 512, MD, 4, torch.float32),
    ("dec.fc1  dW", "TN", 2048, 512, MD, 4, torch.float32),
    ("dec.fc2  dW", "TN", 512, 2048, MD, 4, torch.float32),
    ("dec.proj dW", "TN", 512, 512, MD, 4, torch.float32),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--atomic", action="store_true", help="weight gradients through the atomic split-K epilogue instead of slabs")
    ap.add_argument("--cfg", type=int, default=-1, help="force block tile: 0=128x128 1=256x128 2=256x256 3=256x256 ping-pong")
    ap.add_argument("--lib", action="store_true", help="time torch.matmul (the vendor GEMM library, bf16 out, no epilogue) on the same shapes: a yardstick, not a product path")
    ap.add_argument("--dbg", type=int, default=0, help="ablation bits: 16 main loop only, 32 no DMA, 64 no MFMA, 128 no fragment reads")
    a = ap.parse_args()
    from csmae_hip.engine import Engine
    import csmae_hip
    csmae_hip.load().csmae_gemm_force_tile(a.cfg | a.dbg if a.cfg >= 0 else a.cfg)
    dev = "cuda"
    tot_ms = tot_fl = 0.0
    for name, lay, M, N, K, epi, odt in CASES:
        if a.only and a.only not in name:
            continue
        ta, tb = lay[0] == "T", lay[1] == "N"
        A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16) * 0.05
        C = torch.zeros(M, N, device=dev, dtype=odt)
        bias = torch.randn(N, device=dev) if epi in (0, 1, 2) else None
        aux = torch.randn(M, N, device=dev).to(odt) if epi in (1, 3) else None
        resid = torch.randn(M, N, device=dev).to(odt) if epi == 2 else None   # (the residual stream: bf16 in throughput mode)
        sk = 1
        if epi == 4:
            sk = a.splitk or Engine._splitk(M, N, K, 128, 64)
        kw = dict(trans_a=ta, trans_b=tb, bias=bias, epilogue=epi, aux=aux, resid=resid, splitk=sk)
        if a.lib:
            Al, Bl = (A.t() if ta else A), (B if tb else B.t())
            Cl = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            run = lambda: torch.matmul(Al, Bl, out=Cl)
        elif epi == 4 and not a.atomic:
            ws = torch.empty(64 << 20, device=dev)
            run = lambda: ops.gemm_dw(A, B, C, ws)
        else:
            run = lambda: ops.gemm(A, B, C, **kw)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        print(f"{name:20s} {lay} M={M:6d} N={N:5d} K={K:6d} splitk={sk:3d}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TF/s")
    print(f"{'sum':20s} {tot_ms * 1e3:9.1f} us  {tot_fl / tot_ms / 1e9:8.1f} TF/s")


if __name__ == "__main__":
    main()
