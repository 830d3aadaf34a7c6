#!/usr/bin/env python3
"""Phase timestamps (shader clock) of one workgroup of the two-pairs-per-wave attention backward (attn_bwd1p2_bf16, the decoder shape).
Needs a library whose attention.o was built with -DATTN_TIMING; usage: CSMAE_LIB_PATH=build/abl/lib_attn_ts.so python tools/attn_phase_timing2.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()
for B, T, H, hd in ((256, 197, 16, 32),):
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
    dout = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, T, device="cuda")
    dqkv = torch.empty_like(qkv)
    ops.attn_fwd(qkv, out, lse, B, T, H, hd)
    for _ in range(3):
        ops.attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd)
    e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.csmae_debug_attn_ts(buf)
    ts = list(buf)
    names = {1: "loads issued, lse / delta rows", 2: "staged, K^T fragments", 3: "barrier", 5: "first step", 4: "all steps", 10: "dK / dV stored", 11: "dQ written"}
    print(f"B={B} T={T} H={H} hd={hd}: {e0.elapsed_time(e1) * 100:.1f} us per launch")
    for i in (1, 2, 3, 5, 4, 10, 11):
        print(f"  {names[i]:32s} +{ts[i] - ts[0]:7d} clk")
