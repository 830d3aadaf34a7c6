#!/usr/bin/env python3
"""GPU time between the end of backward and the end of the optimizer step, un-profiled (two events around opt.step()): the AdamW launches
take ~0.57 ms; anything beyond that is idle chip between them (is the host ahead of the GPU at the end of a step?)."""
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
evs = []
def step(rec):
    opt.zero_grad(set_to_none=True)
    loss, _, _ = wrapped(x, mask_ratio=0.75)
    loss.backward()
    if rec:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); opt.step(); b.record(); evs.append((a, b))
    else:
        opt.step()
for _ in range(15):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step(True)
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms; backward-end -> optimizer-end on the GPU: " + " ".join(f"{a.elapsed_time(b) * 1e3:.0f}" for a, b in evs[5:]) + " us")
