#!/usr/bin/env python3
"""GPU time of forward / backward / optimizer of the bench step (HIP events on the main stream, no profiler)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import csmae_hip
csmae_hip.load()
dev = torch.device("cuda", 0)
model, wrapped, opt = bench.build(dev, 128, 1)
x = torch.randn(128, 3, 224, 224, device=dev)
def step(ev=None):
    opt.zero_grad(set_to_none=True)
    if ev: ev[0].record()
    loss, _, _ = wrapped(x, mask_ratio=0.75)
    if ev: ev[1].record()
    loss.backward()
    if ev: ev[2].record()
    opt.step()
    if ev: ev[3].record()
for _ in range(15):
    step()
torch.cuda.synchronize()
tot = [0.0, 0.0, 0.0]
n = 20
for _ in range(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    step(ev)
    torch.cuda.synchronize()
    for k in range(3):
        tot[k] += ev[k].elapsed_time(ev[k + 1])
print("forward %.3f ms  backward %.3f ms  optimizer %.3f ms  sum %.3f" % (tot[0] / n, tot[1] / n, tot[2] / n, sum(tot) / n))
