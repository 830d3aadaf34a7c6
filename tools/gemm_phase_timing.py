#!/usr/bin/env python3
"""Phase timestamps (shader clock) of one workgroup of the pipelined GEMM on a step shape.  Needs a library built with
-DGEMM_TIMING (hipcc ... -DGEMM_TIMING -c csrc/gemm.hip, linked into build/lib_gts.so); usage: gemm_phase_timing.py build/lib_gts.so"""
import ctypes, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
LIB = os.path.join(ROOT, "cross-scale-mae_amd/csmae_hip/libcsmae_hip.so")
shutil.copy(sys.argv[1], LIB)
import torch
from csmae_hip import ops
L = ctypes.CDLL(LIB)
for name, M, N, K, epi in (("dec.qkv fwd", 50432, 1536, 512, 0), ("dec.fc1+gelu fwd", 50432, 2048, 512, 1), ("dec.fc2 fwd (resid)", 50432, 512, 2048, 2),
                           ("enc.qkv fwd", 12800, 2304, 768, 0)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    resid = torch.randn(M, N, device="cuda") if epi == 2 else None
    for _ in range(3):
        ops.gemm(A, B, C, bias=bias, epilogue=epi, aux=aux, resid=resid)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    print(f"{name:22s} prologue {t[1] - t[0]:7d} clk | main loop {t[2] - t[1]:7d} clk ({(t[2] - t[1]) // (K // 64)} per K step) | epilogue {t[3] - t[2]:7d} clk")
