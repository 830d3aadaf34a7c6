#!/usr/bin/env python3
"""hipGraph probe: the bench step (zero_grad -> 2-view forward -> 4 losses -> backward -> fused AdamW) captured
ONCE with torch.cuda.graph and replayed, against the same step enqueued eagerly.  A measurement tool, not a product path: during replay
the crop box, the learning rate and AdamW's bias-correction step are the captured ones (kernel arguments), only the masking noise is
re-drawn (torch's graph-safe Philox).  usage: graph_probe.py [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import csmae_hip
csmae_hip.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
model, wrapped, opt = bench.build(dev, 128, 1)
x = torch.randn(128, 3, 224, 224, device=dev)
# the crop box reaches the device by an H2D copy from pageable memory (not capturable): give the draw a pinned buffer
pin = torch.zeros(4, dtype=torch.int32).pin_memory()
draw = model._draw
def draw_pinned(*a, **k):
    noise, box = draw(*a, **k)
    pin.copy_(box)
    return noise, pin
model._draw = draw_pinned

def step():
    opt.zero_grad(set_to_none=True)
    loss, _, _ = wrapped(x, mask_ratio=0.75)
    loss.backward()
    opt.step()
    return loss

def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(12):
    step()
eager = [timed(step, K) for _ in range(3)]
print(f"eager  {min(eager):.3f} ms/step (runs: {', '.join(f'{v:.3f}' for v in eager)})", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
print(f"captured in {time.perf_counter() - t0:.2f} s", flush=True)
g.replay(); torch.cuda.synchronize()
print(f"loss after a replay: {float(static_loss):.5f}")
graph = [timed(g.replay, K) for _ in range(3)]
print(f"graph  {min(graph):.3f} ms/step (runs: {', '.join(f'{v:.3f}' for v in graph)})")
eager2 = [timed(step, K) for _ in range(2)]
print(f"eager  {min(eager2):.3f} ms/step again (runs: {', '.join(f'{v:.3f}' for v in eager2)})")
