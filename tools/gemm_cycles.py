#!/usr/bin/env python3
"""Shader clocks per K step of one workgroup (block 300, wave 0) of a GEMM variant library built with -DGEMM_TIMING.
usage: gemm_cycles.py CFG name...   (libraries build/abl/libcsmae_pp_<name>.so, one process per library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] != "--one":
    for n in sys.argv[2:]:
        subprocess.run([sys.executable, __file__, "--one", sys.argv[1], n])
    sys.exit(0)
cfg, name = int(sys.argv[2]), sys.argv[3]
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, f"build/abl/libcsmae_pp_{name}.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()
L.csmae_gemm_force_tile(cfg)
out = []
for label, M, N, K in (("dec.qkv", 50432, 1536, 512), ("enc.qkv", 12800 * 2, 2304, 768), ("fc2", 12800 * 3, 768, 3072)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(A, B, C)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    out.append(f"{label}: prologue {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} = {(t[2] - t[1]) / (K // 64):6.0f}/step epilogue {t[3] - t[2]:5d}")
print(f"cfg {cfg} {name:8s} " + " | ".join(out))
