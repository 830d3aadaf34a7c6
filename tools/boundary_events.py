#!/usr/bin/env python3
"""Main-stream time between the end of a step's backward pass and the first kernels of the next forward pass, by HIP events (no profiler attached):
backward end -> before the mask noise draw -> after it -> next backward end.  With FusedAdamW(overlap=True) nothing of the optimizer is on the main stream,
so the first interval is pure idle time.   python tools/boundary_events.py [--steps 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, wrapped, opt = bench.build(dev, 128, 1)
    x = torch.randn(128, 3, 224, 224, device=dev)
    E = lambda: torch.cuda.Event(enable_timing=True)
    marks = []

    def step(rec):
        opt.zero_grad(set_to_none=True)
        e0 = E(); e0.record()
        loss, _, _ = wrapped(x, mask_ratio=0.75)
        e1 = E(); e1.record()
        loss.backward()
        e2 = E(); e2.record()
        opt.step()
        e3 = E(); e3.record()
        if rec:
            marks.append((e0, e1, e2, e3))
    for _ in range(15):
        step(False)
    torch.cuda.synchronize()
    for _ in range(a.steps):
        step(True)
    torch.cuda.synchronize()
    fw = [m[0].elapsed_time(m[1]) for m in marks]
    bw = [m[1].elapsed_time(m[2]) for m in marks]
    op = [m[2].elapsed_time(m[3]) for m in marks]
    gap = [marks[i][3].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
    tot = [marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"main stream, medians over {a.steps} steps: forward {med(fw):.3f} ms | backward {med(bw):.3f} ms | optimizer (on this stream) {med(op):.3f} ms | "
          f"step end -> next forward's first kernel {med(gap):.3f} ms | step {med(tot):.3f} ms")


if __name__ == "__main__":
    main()
