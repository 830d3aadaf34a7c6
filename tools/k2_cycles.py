#!/usr/bin/env python3
"""Shader clocks of one workgroup (block 300, wave 0) of the two-workgroups-per-CU GEMM by phase, next to the one-workgroup kernel, from
variant libraries built by tools/k2_variants.sh (-DGEMM_TIMING).  usage: k2_cycles.py name...   (one process per library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] != "--one":
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", n])
    sys.exit(0)
name = sys.argv[2]
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, f"build/abl/libcsmae_k2_{name}.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()


def ts(k2):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    (L.csmae_debug_k2_ts if k2 else L.csmae_debug_gemm_ts)(buf)
    t = list(buf)
    return t


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"== {name}")
for label, M, N, K, epi in (("dec.qkv", 50432, 1536, 512, 0), ("dec.fc1+gelu", 50432, 2048, 512, 1), ("enc.qkv x2", 25600, 2304, 768, 0), ("enc.fc2 x3 resid", 38400, 768, 3072, 2)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Wt = W.t().contiguous()
    Wk = W.view(N, K // 32, 32).permute(1, 0, 2).contiguous().view(-1)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    aux = torch.zeros(M, N, device="cuda", dtype=torch.uint8) if epi == 1 else None
    resid = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == 2 else None
    row = []
    for kind in ("k64 NT", "k2 NT", "k64 NN", "k2 NN"):
        L.csmae_gemm_k2_mode(3 * int(kind.startswith("k2")), 3 * int(kind.startswith("k2")))
        if kind.endswith("NT"):
            fn = (lambda: ops.gemm_ks(A, Wk, W, C, bias=bias, epilogue=epi, aux=aux, resid=resid)) if kind.startswith("k2") else (lambda: ops.gemm(A, W, C, bias=bias, epilogue=epi, aux=aux, resid=resid))
        else:
            fn = lambda: ops.gemm(A, Wt, C, trans_b=True, bias=bias, epilogue=epi, aux=aux, resid=resid)
        us = timed(fn)
        t = ts(kind.startswith('k2'))
        row.append(f"{kind}: {us:6.1f} us  pro {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} = {(t[2] - t[1]) / (K // 64):5.0f}/step epi {t[3] - t[2]:5d}")
    print(f"{label:18s} " + " | ".join(row))
