#!/usr/bin/env python3
"""Host-side enqueue time of each phase of the bench step (no synchronisation inside the step): is the step CPU- or GPU-bound?"""
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
def step(rec=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t1 = time.perf_counter()
    loss, _, _ = wrapped(x, mask_ratio=0.75)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if rec is not None:
        rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(12):
    step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(20):
    step(rec)
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
import statistics
names = ["zero_grad", "forward", "backward", "opt.step"]
for i, n in enumerate(names):
    print(f"{n:10s} host {statistics.mean(r[i] for r in rec) * 1e3:7.3f} ms/step")
print(f"host enqueue total {t_cpu / 20 * 1e3:.3f} ms/step; wall incl. drain {t_all / 20 * 1e3:.3f} ms/step")
