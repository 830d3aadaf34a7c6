#!/usr/bin/env python3
"""Yardstick: the vendor libraries on the step's shapes next to this repo's kernels, same box, same HIP events.

    python tools/gemm_yardstick.py [--iters 20] > profiles/r05_yardstick.txt

A TOOL — never imported by the product path.  `torch.matmul` (hipBLASLt / rocBLAS behind ATen) computes the bare bf16 product
(bf16 out, no bias / GELU / residual / bias-gradient epilogue), `torch.nn.functional.scaled_dot_product_attention` the bare attention;
`csmae_gemm` / `csmae_attn_*` run WITH their fused epilogues.  The point is to learn what gfx950 reaches on these shapes, not to race.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    print(f"# {torch.cuda.get_device_name(0)}  torch {torch.__version__}  (us per launch, alone on the chip; TF/s = 2MNK / time)")
    print(f"{'product':20s} {'lay':3s} {'M':>6s} {'N':>5s} {'K':>6s} | {'csmae us':>9s} {'TF/s':>7s} | {'vendor us':>9s} {'TF/s':>7s} | csmae/vendor time")
    tot = [0.0, 0.0, 0.0]
    ws = torch.empty(64 << 20, device=dev)
    for name, lay, M, N, K, epi, odt in CASES:
        ta, tb = lay[0] == "T", lay[1] == "N"
        A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16) * 0.05
        C = torch.zeros(M, N, device=dev, dtype=odt)
        bias = torch.randn(N, device=dev) if epi in (0, 1, 2) else None
        aux = torch.randint(0, 255, (M, N), device=dev, dtype=torch.uint8) if epi in (1, 3) else None   # 8-bit gelu' codes, as in the step
        resid = torch.randn(M, N, device=dev).to(odt) if epi == 2 else None
        if epi == 4:
            ours = lambda: ops.gemm_dw(A, B, C, ws)
        else:
            ours = lambda: ops.gemm(A, B, C, trans_a=ta, trans_b=tb, bias=bias, epilogue=epi, aux=aux, resid=resid)
        Al, Bl = (A.t() if ta else A), (B if tb else B.t())
        Cl = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        lib = lambda: torch.matmul(Al, Bl, out=Cl)
        t0, t1 = timed(ours, a.iters), timed(lib, a.iters)
        fl = 2.0 * M * N * K
        tot[0] += t0; tot[1] += t1; tot[2] += fl
        print(f"{name:20s} {lay:3s} {M:6d} {N:5d} {K:6d} | {t0 * 1e3:9.1f} {fl / t0 / 1e9:7.1f} | {t1 * 1e3:9.1f} {fl / t1 / 1e9:7.1f} | {t0 / t1:5.2f}")
    print(f"{'sum (20+4 shapes)':44s} | {tot[0] * 1e3:9.1f} {tot[2] / tot[0] / 1e9:7.1f} | {tot[1] * 1e3:9.1f} {tot[2] / tot[1] / 1e9:7.1f} | {tot[0] / tot[1]:5.2f}")
    print()
    print("# attention: csmae_attn_fwd / csmae_attn_bwd on packed qkv [B*T, 3*H*hd] vs torch SDPA on [B, H, T, hd] (forward; backward = autograd of it)")
    for name, Bn, T, H, hd in [("dec", 256, 197, 16, 32), ("enc", 256, 50, 12, 64)]:
        D = H * hd
        qkv = torch.randn(Bn * T, 3 * D, device=dev).to(torch.bfloat16)
        dout = torch.randn(Bn * T, D, device=dev).to(torch.bfloat16)
        out = torch.empty(Bn * T, D, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(Bn, H, T, device=dev)
        dqkv = torch.empty_like(qkv)
        tf = timed(lambda: ops.attn_fwd(qkv, out, lse, Bn, T, H, hd), a.iters)
        tb_ = timed(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, Bn, T, H, hd), a.iters)
        q, k, v = (torch.randn(Bn, H, T, hd, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        g = torch.randn(Bn, H, T, hd, device=dev, dtype=torch.bfloat16)
        sf = timed(lambda: F.scaled_dot_product_attention(q, k, v), a.iters)

        def fb():
            o = F.scaled_dot_product_attention(q, k, v)
            o.backward(g)
            q.grad = k.grad = v.grad = None
        sfb = timed(fb, a.iters)
        print(f"attn {name} B={Bn} T={T} H={H} hd={hd}: csmae fwd {tf * 1e3:7.1f} us  bwd {tb_ * 1e3:7.1f} us | SDPA fwd {sf * 1e3:7.1f} us  fwd+bwd {sfb * 1e3:7.1f} us (bwd ~ {max(sfb - sf, 0) * 1e3:7.1f})")


if __name__ == "__main__":
    main()
