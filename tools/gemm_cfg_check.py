#!/usr/bin/env python3
"""Compare a forced GEMM tile configuration against the default one on odd shapes (tuning aid)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import csmae_hip
from csmae_hip import ops

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = csmae_hip.load()
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(256, 256, 32), (256, 256, 64), (300, 260, 96), (1000, 520, 160), (777, 512, 1000), (2048, 768, 3072), (513, 1028, 40)]:
    for ta, tb in [(False, False), (False, True), (True, True), (True, False)]:
        if not (ta and tb) and K % 8:
            continue
        Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8
        A = torch.randn((K, Mp) if ta else (M, K), device="cuda").to(torch.bfloat16)
        B = torch.randn((K, Np) if tb else (N, K), device="cuda").to(torch.bfloat16)
        if ta: A = A[:, :M]
        if tb: B = B[:, :N]
        outs = []
        for c in (2, cfg):
            lib.csmae_gemm_force_tile(c)
            C = torch.zeros(M, N, device="cuda", dtype=torch.float32)
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb)
            outs.append(C)
        lib.csmae_gemm_force_tile(-1)
        d = (outs[0] - outs[1]).abs().max().item()
        ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
        e = (outs[1] - ref).abs().max().item() / ref.abs().max().item()
        flag = "" if d == 0 and e < 1e-2 else "  <-- MISMATCH"
        bad += bool(flag)
        print(f"M={M} N={N} K={K} ta={ta} tb={tb} max|cfg{cfg}-cfg2|={d:.3g} rel_err_vs_fp32={e:.2e}{flag}")
print("FAILED" if bad else "OK")
