#!/usr/bin/env python3
"""Shader clocks of one workgroup (block 300) of the plain and the LayerNorm-folded forward products, from a -DGEMM_TIMING build
(tools/gemm_variants.sh timing="-DGEMM_TIMING"): prologue | main loop | epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, "build/abl/libcsmae_pp_timing.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import EPI_GELU, EPI_RESID, ops
L = csmae_hip.load()


def ts():
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    return f"prologue {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} epilogue {t[3] - t[2]:6d} total {t[3] - t[0]:6d}"


for label, M, D in (("dec", 25216, 512), ("enc", 6400 * 2, 768)):
    parts = (D + 255) // 256
    x = (torch.randn(M, D, device="cuda") * 1.5).to(torch.bfloat16)
    st = torch.rand(parts, M, 2, device="cuda") + 300.0
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    for name, N, gelu in (("qkv", 3 * D, False), ("fc1", 4 * D, True)):
        w = (torch.randn(N, D, device="cuda") * D ** -0.5).to(torch.bfloat16)
        b, c = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        aux = torch.empty(M, N, device="cuda", dtype=torch.uint8) if gelu else None
        epi = EPI_GELU if gelu else 0
        for _ in range(3):
            ops.gemm(x, w, out, bias=b, epilogue=epi, aux=aux)
        print(f"{label}.{name} plain : {ts()}")
        for _ in range(3):
            ops.gemm_lnfold(x, w, out, c, b, st, parts, mean, rstd, epilogue=epi, aux=aux)
        print(f"{label}.{name} folded: {ts()}")
    for name, K in (("proj", D), ("fc2", 4 * D)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(D, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(D, device="cuda")
        out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(a, w, out, bias=b, epilogue=EPI_RESID, resid=x)
        print(f"{label}.{name} plain : {ts()}")
        for _ in range(3):
            ops.gemm_resid_stats(a, w, out, b, x, st)
        print(f"{label}.{name} +stats: {ts()}")
