#!/usr/bin/env python3
"""Per-step wall time of the bench workload (one sync per step) to look at run-to-run / step-to-step jitter."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import csmae_hip
csmae_hip.load()
dev = torch.device("cuda", 0)
model, wrapped, opt = bench.build(dev, 128, 1)
x = torch.randn(128, 3, 224, 224, device=dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss, _, _ = wrapped(x, mask_ratio=0.75)
    loss.backward()
    opt.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))
