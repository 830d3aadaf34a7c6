#!/usr/bin/env python3
"""Shader clocks of one workgroup (block GTS_BLOCK, wave 0) of the grouped weight-gradient kernel by phase (a -DGEMM_TIMING -DGTS_BLOCK=100 build:
tools/k2_variants.sh dw100="-DGTS_BLOCK=100" builds gemm.hip with it when GEMM_EXTRA is set).  usage: dw_cycles.py LIB"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CSMAE_LIB_PATH"] = os.path.join(ROOT, sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()


def run(label, K, prods, slots, mode=0):
    dev = "cuda"
    L.csmae_gemm_dw_mode(mode)
    ws = torch.empty(96 << 20, device=dev)
    items = []
    for M, N in prods:
        dy = torch.randn(K, M, device=dev).to(torch.bfloat16)
        x = torch.randn(K, N, device=dev).to(torch.bfloat16)
        items.append((dy, x, torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)))
    grp = ops.DwGroup(items, ws)
    for _ in range(3):
        grp.launch(slots)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        grp.launch(slots)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    buf = (ctypes.c_ulonglong * 8)()
    (L.csmae_debug_k2_ts if mode else L.csmae_debug_gemm_ts)(buf)
    t = list(buf)
    tiles = sum(-(-M // (128 if mode else 256)) * -(-N // 256) for M, N in prods)
    S = max(1, min((2 * slots if mode else slots) // tiles, (K // 64) // 8))
    kps = -(-(K // 64) // S)
    fl = sum(2.0 * M * N * K for M, N in prods)
    print(f"{'k2 ' if mode else 'k64'} {label:28s} slots {slots:3d}: {us:7.1f} us {fl / us / 1e6:7.1f} TF/s | tiles {tiles} x {S} slices of {kps} steps: pro {t[1] - t[0]:6d} loop {t[2] - t[1]:7d} = {(t[2] - t[1]) / kps:5.0f}/step epi {t[3] - t[2]:6d}")


for slots in (128, 160, 256):
    for mode in (0, 1):
        run("dec fc2+fc1 (K=50432)", 50432, [(512, 2048), (2048, 512)], slots, mode)
        run("dec proj+qkv (K=50432)", 50432, [(512, 512), (1536, 512)], slots, mode)
        run("enc fc2+fc1 (K=12800)", 12800, [(768, 3072), (3072, 768)], slots, mode)
        run("enc proj+qkv (K=12800)", 12800, [(768, 768), (2304, 768)], slots, mode)
