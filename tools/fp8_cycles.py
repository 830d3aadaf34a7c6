#!/usr/bin/env python3
"""Shader clocks of one workgroup (block 300) of the pipelined fp8 product on ViT-H/14 shapes, from a -DGEMM_TIMING build."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, "build/abl/libcsmae_pp_timing.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import EPI_GELU, EPI_RESID, ops
L = csmae_hip.load()
for label, M, N, K, epi in (("enc.qkv", 33280, 3840, 1280, 0), ("enc.fc1", 33280, 5120, 1280, EPI_GELU), ("enc.fc2", 33280, 1280, 5120, EPI_RESID), ("dec.fc1", 131584, 2048, 512, EPI_GELU)):
    a8 = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8)
    w8 = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8)
    dq = torch.ones(1, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.empty(M, N, device="cuda", dtype=torch.uint8) if epi == EPI_GELU else None
    resid = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == EPI_RESID else None
    bias = torch.zeros(N, device="cuda")
    for _ in range(3):
        ops.gemm_fp8(a8, w8, out, dq, dq, bias=bias, epilogue=epi, aux=aux, resid=resid)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_fp8(a8, w8, out, dq, dq, bias=bias, epilogue=epi, aux=aux, resid=resid)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{label} {M}x{N}x{K}: prologue {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} = {(t[2] - t[1]) / (K // 128):6.0f}/step epilogue {t[3] - t[2]:6d} | {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.0f} TF")
