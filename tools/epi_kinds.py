#!/usr/bin/env python3
"""Phase clocks (one workgroup, -DGEMM_TIMING build `timing`) of the four epilogue kinds on decoder shapes; run twice, with and without
CSMAE_EPI_POINTERS=1, to compare the buffer-addressed row-segment epilogue with the pointer-addressed one."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, "build/abl/libcsmae_pp_timing.so")
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import EPI_DGELU, EPI_GELU, EPI_RESID, ops
L = csmae_hip.load()
M, D = 50432, 512


def ts(label):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L.csmae_debug_gemm_ts(buf)
    t = list(buf)
    print(f"{label:22s} prologue {t[1] - t[0]:5d} loop {t[2] - t[1]:6d} epilogue {t[3] - t[2]:6d}")


bf = dict(device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, D, **bf)
w1 = (torch.randn(4 * D, D, device="cuda") * D ** -0.5).to(torch.bfloat16)
w2 = (torch.randn(D, 4 * D, device="cuda") * (4 * D) ** -0.5).to(torch.bfloat16)
b1, b2 = torch.randn(4 * D, device="cuda"), torch.randn(D, device="cuda")
h = torch.empty(M, 4 * D, **bf)
gq = torch.empty(M, 4 * D, device="cuda", dtype=torch.uint8)
out = torch.empty(M, D, **bf)
for _ in range(3):
    ops.gemm(x, w1, h, bias=b1, epilogue=EPI_GELU, aux=gq)
ts("NT fc1 GELU+q8")
for _ in range(3):
    ops.gemm(h, w2, out, bias=b2, epilogue=EPI_RESID, resid=x)
ts("NT fc2 RESID")
dpre = torch.empty(M, 4 * D, **bf)
for _ in range(3):
    ops.gemm(out, w2, dpre, trans_b=True, epilogue=EPI_DGELU, aux=gq)
ts("NN fc2-dX DGELU q8")
for _ in range(3):
    ops.gemm(dpre, w1, out, trans_b=True)
ts("NN fc1-dX plain")
