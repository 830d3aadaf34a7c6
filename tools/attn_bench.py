#!/usr/bin/env python3
"""Attention fwd/bwd micro-benchmark on the step's shapes (decoder: B=256,T=197,H=16,hd=32; encoder: B=256,T=50,H=12,hd=64; --huge14: the ViT-H/14 preset's)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
from csmae_hip import ops  # noqa: E402

SHAPES = [("dec", 256, 197, 16, 32), ("enc", 256, 50, 12, 64)]
if "--huge14" in sys.argv:   # ViT-H/14 at 256 per GPU (BASELINE.json configs[4]): encoder 65 tokens x 16 heads of 80, decoder 257 x 16 x 32
    sys.argv.remove("--huge14")
    SHAPES = [("h14 enc", 512, 65, 16, 80), ("h14 dec", 512, 257, 16, 32)]
for name, B, T, H, hd in SHAPES:
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
    dout = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device="cuda")
    dqkv = torch.empty_like(qkv)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for fn, label, fl in ((lambda: ops.attn_fwd(qkv, out, lse, B, T, H, hd), "fwd", 4.0), (lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd), "bwd", 10.0)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"attn {name} {label}: {ms * 1e3:8.1f} us  {fl * B * H * T * T * hd / ms / 1e9:7.1f} TF/s (algorithmic)")
