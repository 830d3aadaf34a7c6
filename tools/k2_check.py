#!/usr/bin/env python3
"""Correctness + timing of the two-workgroups-per-CU GEMM (csrc/gemm_k2.hip) against the one-workgroup kernels on the step's shapes.
    python tools/k2_check.py [--iters 20] [--quick]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import csmae_hip  # noqa: E402
from csmae_hip import ops  # noqa: E402
from gemm_bench import CASES  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kslab(w):
    N, K = w.shape
    desc = torch.tensor([[0, N, K]], dtype=torch.long, device=w.device)
    dst = torch.zeros(N * K, device=w.device, dtype=torch.bfloat16)
    ops.weights_kslab(desc, w.reshape(-1), dst)
    ref = w.view(N, K // 32, 32).permute(1, 0, 2).contiguous().view(-1)
    assert torch.equal(dst, ref), "csmae_weights_kslab layout"
    return dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    lib = csmae_hip.load()
    dev = "cuda"
    torch.manual_seed(0)
    # ---- small / ragged correctness cases against fp32 torch
    small = [(128, 256, 64), (130, 256, 128), (300, 264, 192), (1000, 768, 768), (257, 512, 512), (128, 1024, 64), (515, 2304, 768)]
    for M, N, K in small:
        for lay in ("NT", "NN"):
            A = torch.randn(M, K, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            bias = torch.randn(N, device=dev)
            ref = A.float() @ W.float().t() + bias
            C2 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            C1 = torch.empty_like(C2)
            if lay == "NT":
                Wk = kslab(W)
                lib.csmae_gemm_k2_mode(3, 3)
                ops.gemm_ks(A, Wk, W, C2, bias=bias)
                ops.gemm(A, W, C1, bias=bias)
            else:
                Wt = W.t().contiguous()   # [K][N]
                lib.csmae_gemm_k2_mode(3, 3)
                ops.gemm(A, Wt, C2, trans_b=True, bias=bias)
                lib.csmae_gemm_k2_mode(0, 0)
                ops.gemm(A, Wt, C1, trans_b=True, bias=bias)
                lib.csmae_gemm_k2_mode(3, 3)
            torch.cuda.synchronize()
            err = (C2.float() - ref).abs().max().item() / ref.abs().max().item()
            same = torch.equal(C1, C2)
            print(f"check {lay} M={M} N={N} K={K}: rel err vs fp32 {err:.2e}  bit-identical to the one-workgroup kernel: {same}")
            assert err < 1e-2, "k2 kernel wrong"
    if a.quick:
        return
    # ---- the step's shapes: epilogues, bit-identity, timing
    tot = [0.0, 0.0, 0.0]
    print(f"{'product':20s} {'lay':3s} {'M':>6s} {'N':>5s} {'K':>6s} | {'k64 us':>8s} {'TF/s':>7s} | {'k2 us':>8s} {'TF/s':>7s} | k2/k64  same")
    for name, lay, M, N, K, epi, odt in CASES:
        if lay == "TN":
            continue
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) if epi in (0, 1, 2) else None
        resid = torch.randn(M, N, device=dev).to(odt) if epi == 2 else None
        aux1 = torch.zeros(M, N, device=dev, dtype=torch.uint8) if epi in (1, 3) else None
        aux2 = torch.zeros(M, N, device=dev, dtype=torch.uint8) if epi in (1, 3) else None
        if epi == 3:
            aux1.random_(0, 255); aux2.copy_(aux1)
        C1 = torch.zeros(M, N, device=dev, dtype=odt)
        C2 = torch.zeros(M, N, device=dev, dtype=odt)
        if lay == "NT":
            Wk = kslab(W)
            f1 = lambda: ops.gemm(A, W, C1, bias=bias, epilogue=epi, aux=aux1, resid=resid)
            f2 = lambda: ops.gemm_ks(A, Wk, W, C2, bias=bias, epilogue=epi, aux=aux2, resid=resid)
            lib.csmae_gemm_k2_mode(3, 3)
            t1, t2 = timed(f1, a.iters), timed(f2, a.iters)
        else:
            Wt = W.t().contiguous()
            f1 = lambda: ops.gemm(A, Wt, C1, trans_b=True, bias=bias, epilogue=epi, aux=aux1, resid=resid)
            f2 = lambda: ops.gemm(A, Wt, C2, trans_b=True, bias=bias, epilogue=epi, aux=aux2, resid=resid)
            lib.csmae_gemm_k2_mode(0, 0)
            t1 = timed(f1, a.iters)
            lib.csmae_gemm_k2_mode(3, 3)
            t2 = timed(f2, a.iters)
        torch.cuda.synchronize()
        same = torch.equal(C1, C2) and (aux1 is None or torch.equal(aux1, aux2))
        fl = 2.0 * M * N * K
        tot[0] += t1; tot[1] += t2; tot[2] += fl
        print(f"{name:20s} {lay:3s} {M:6d} {N:5d} {K:6d} | {t1 * 1e3:8.1f} {fl / t1 / 1e9:7.1f} | {t2 * 1e3:8.1f} {fl / t2 / 1e9:7.1f} | {t2 / t1:5.2f}  {same}")
    print(f"{'sum':44s} | {tot[0] * 1e3:8.1f} {tot[2] / tot[0] / 1e9:7.1f} | {tot[1] * 1e3:8.1f} {tot[2] / tot[1] / 1e9:7.1f} | {tot[1] / tot[0]:5.2f}")


if __name__ == "__main__":
    main()
