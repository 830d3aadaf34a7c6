#!/usr/bin/env python3
"""Is the GEMM epilogue bound by the CU's own store path or by the chip's write bandwidth?  Times block 300 of a -DGEMM_TIMING build when
it runs in a full round (every CU in its epilogue at the same time) and in a nearly empty second round.  usage: epi_probe.py [libname]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "g0"
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, f"build/abl/libcsmae_pp_{name}.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()
L.csmae_gemm_force_tile(4)
for label, M, N, K, epi in (("full rounds", 50432, 1536, 512, 0), ("block 300 in a 56-tile last round", 13312, 1536, 512, 0), ("full rounds, gelu", 50432, 2048, 512, 1),
                            ("56-tile last round, gelu", 256 * 39, 2048, 512, 1), ("full, K=2048 resid", 50432, 512, 2048, 2), ("last round, K=2048 resid", 256 * 156, 512, 2048, 2)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    resid = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 2 else None
    for _ in range(5):
        ops.gemm(A, B, C, bias=bias, epilogue=epi, aux=aux, resid=resid)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    print(f"{label:36s} tiles {((M + 255) // 256) * ((N + 255) // 256):5d}: prologue {t[1] - t[0]:6d} loop {t[2] - t[1]:7d} ({(t[2] - t[1]) // (K // 64)}/step) epilogue {t[3] - t[2]:6d}")
