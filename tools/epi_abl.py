#!/usr/bin/env python3
"""Where the clocks of the fc1 (GELU + 8-bit gelu') epilogue go: phase clocks of one workgroup for -DGEMM_TIMING builds with parts of the
epilogue compiled out (-DGEMM_EPI_ABL: 1 no GELU arithmetic, 2 no gelu' store, 4 no C store).  usage: epi_abl.py name..."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "--one", n])
    sys.exit(0)
name = sys.argv[2]
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, f"build/abl/libcsmae_pp_{name}.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import EPI_GELU, EPI_RESID, ops
L = csmae_hip.load()
out = []
for label, M, N, K in (("dec.fc1", 25216, 2048, 512), ("enc.fc1", 12800, 3072, 768)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    for _ in range(3):
        ops.gemm(a, w, h, bias=b, epilogue=EPI_GELU, aux=aux)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    out.append(f"{label}: loop {t[2] - t[1]:6d} epilogue {t[3] - t[2]:6d}")
print(f"{name:10s} " + " | ".join(out))
