#!/usr/bin/env python3
"""Attention kernel time vs number of workgroups (decoder shape) to separate per-block latency from throughput limits."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import ops
T, H, hd = 197, 16, 32
D = H * hd
for B in (1, 4, 8, 16, 32, 64, 128, 256):
    qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
    dout = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device="cuda")
    dqkv = torch.empty_like(qkv)
    res = []
    for fn in (lambda: ops.attn_fwd(qkv, out, lse, B, T, H, hd), lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd)):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"B={B:4d} blocks={B * H:5d}  fwd {res[0]:8.1f} us  bwd {res[1]:8.1f} us")
