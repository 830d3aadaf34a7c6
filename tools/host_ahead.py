#!/usr/bin/env python3
"""How far the host runs ahead of the GPU in the bench step: host time per phase of enqueueing one step (zero_grad / forward / backward / optimizer)
against the step's GPU time.  A host that is not ahead at a step boundary shows as GPU idle there (tools/step_timeline.py).
    python tools/host_ahead.py [--steps 30]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, wrapped, opt = bench.build(dev, 128, 1)
    x = torch.randn(128, 3, 224, 224, device=dev)

    def step(t):
        t.append(time.perf_counter()); opt.zero_grad(set_to_none=True)
        t.append(time.perf_counter()); loss, _, _ = wrapped(x, mask_ratio=0.75)
        t.append(time.perf_counter()); loss.backward()
        t.append(time.perf_counter()); opt.step()
        t.append(time.perf_counter())
    for _ in range(15):
        step([])
    torch.cuda.synchronize()
    rows, evs, ahead = [], [], []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t = []
        step(t)
        rows.append(t)
        evs.append(torch.cuda.Event())
        evs[-1].record()
        ahead.append(sum(not e.query() for e in evs))   # steps enqueued whose main-stream work the GPU has not finished
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    ph = [sum(r[i + 1] - r[i] for r in rows) / len(rows) * 1e3 for i in range(4)]
    print(f"{a.steps} steps: host enqueue {t_enq / a.steps * 1e3:.2f} ms/step, wall {t_all / a.steps * 1e3:.2f} ms/step; host per phase: zero_grad {ph[0]:.3f}  forward {ph[1]:.3f}  "
          f"backward {ph[2]:.3f}  optimizer {ph[3]:.3f} ms")
    print("  steps the host is ahead of the GPU after enqueueing each step:", ahead)
    # the first few steps after a synchronisation show the host's own pace (nothing to wait for)
    for k, r in enumerate(rows[:3]):
        print(f"  step {k}: " + "  ".join(f"{(r[i + 1] - r[i]) * 1e3:.3f}" for i in range(4)))


if __name__ == "__main__":
    main()
