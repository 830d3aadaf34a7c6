#!/usr/bin/env python3
"""Phase timestamps (shader clock) of one workgroup of the single-pass attention backward.  Needs a library built with
-DATTN_TIMING (hipcc ... -DATTN_TIMING -c csrc/attention.hip, linked into build/lib_ts.so); usage: attn_phase_timing.py build/lib_ts.so"""
import ctypes, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
LIB = os.path.join(ROOT, "cross-scale-mae_amd/csmae_hip/libcsmae_hip.so")
shutil.copy(sys.argv[1], LIB)
import torch
from csmae_hip import ops
B, T, H, hd = 256, 197, 16, 32
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
dout = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
out = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_bwd(qkv, out, dout, lse, dqkv, B, T, H, hd)
torch.cuda.synchronize()
L = ctypes.CDLL(LIB)
buf = (ctypes.c_ulonglong * 16)()
print("rc", L.csmae_debug_attn_ts(buf))
ts = list(buf)
names = {0: "start", 1: "loads issued + D rows done", 2: "sweep 0 K/V fragments", 3: "sweep 0 barrier passed", 4: "sweep 0 loop done",
         6: "sweep 1 K/V fragments", 7: "sweep 1 barrier passed", 8: "sweep 1 loop done", 10: "before dQ write-out"}
for i in sorted(names):
    print(f"{names[i]:28s} +{(ts[i] - ts[0]):8d} clk")
