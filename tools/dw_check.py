#!/usr/bin/env python3
"""Grouped weight gradients: the two-workgroups-per-CU kernel (k2_tile_tn) against the one-workgroup kernel and fp32 torch; timing on the step's groups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-scale-mae_amd"))
import torch
import csmae_hip
from csmae_hip import ops
L = csmae_hip.load()
dev = "cuda"
torch.manual_seed(0)


def group(K, prods, pad=0):
    items = []
    for M, N in prods:
        dy = torch.randn(K, M + pad, device=dev).to(torch.bfloat16)[:, :M]
        x = torch.randn(K, N + pad, device=dev).to(torch.bfloat16)[:, :N]
        items.append((dy, x))
    return items


def run(items, slots, mode, ws, init=0.5):
    L.csmae_gemm_dw_mode(mode)
    outs = [(torch.full((dy.shape[1], x.shape[1]), init, device=dev), torch.full((dy.shape[1],), init, device=dev)) for dy, x in items]
    grp = ops.DwGroup([(dy, x, dw, db) for (dy, x), (dw, db) in zip(items, outs)], ws)
    grp.launch(slots)
    torch.cuda.synchronize()
    return outs, grp


ws = torch.empty(128 << 20, device=dev)
for K, prods, slots, pad in ((12800, [(768, 3072), (3072, 768)], 256, 0), (512, [(256, 256)], 4, 0), (1000, [(256, 512), (520, 264)], 16, 8), (4096, [(768, 768), (2304, 768)], 160, 0), (3000, [(512, 2048), (2048, 512)], 128, 0),
                             (12800, [(768, 3072), (3072, 768)], 160, 0), (640, [(1024, 1024), (3072, 1024), (1024, 4096), (4096, 1024)], 256, 0)):
    items = group(K, prods, pad)
    o2, _ = run(items, slots, 1, ws)
    o2b, _ = run(items, slots, 1, ws)
    o1, _ = run(items, slots, 0, ws)
    for (dy, x), (dw2, db2), (dw2b, db2b), (dw1, db1) in zip(items, o2, o2b, o1):
        ref = 0.5 + dy.float().t() @ x.float()
        refb = 0.5 + dy.float().sum(0)
        e2 = (dw2 - ref).abs().max().item() / ref.abs().max().item()
        e1 = (dw1 - ref).abs().max().item() / ref.abs().max().item()
        eb = (db2 - refb).abs().max().item() / refb.abs().max().item()
        print(f"K={K} {tuple(dw2.shape)} slots {slots}: k2 rel err {e2:.2e} (one-workgroup {e1:.2e}) db {eb:.2e} deterministic {torch.equal(dw2, dw2b) and torch.equal(db2, db2b)}")
        assert e2 < 2e-3 and eb < 2e-3 and torch.equal(dw2, dw2b) and torch.equal(db2, db2b)
if len(sys.argv) > 1 and sys.argv[1] == "--quick":
    sys.exit(0)
print("timing (us per launch incl. fold):")
for label, K, prods in (("dec fc2+fc1", 50432, [(512, 2048), (2048, 512)]), ("dec proj+qkv", 50432, [(512, 512), (1536, 512)]), ("enc fc2+fc1", 12800, [(768, 3072), (3072, 768)]), ("enc proj+qkv", 12800, [(768, 768), (2304, 768)])):
    items = group(K, prods)
    fl = sum(2.0 * dy.shape[1] * x.shape[1] * K for dy, x in items)
    row = []
    for slots in (128, 160, 256):
        for mode in (0, 1):
            _, grp = run(items, slots, mode, ws)
            for _ in range(3):
                grp.launch(slots)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                grp.launch(slots)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            row.append(f"{'k2' if mode else 'k64'}@{slots}: {us:6.1f} us {fl / us / 1e6:6.0f} TF")
    print(f"{label:14s} " + " | ".join(row))
