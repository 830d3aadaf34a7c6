#!/usr/bin/env python3
"""Per-stream timeline of one training step from a rocprofv3 rocpd DB: busy time per stream, idle gaps on the main stream,
and a phase table (forward / backward / optimizer).  usage: step_timeline.py results.db [step_index]"""
import sqlite3
import sys


def main(db, step=-2):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    starts = [r[1] for r in rows if "crop_resize" in r[0]]
    s0, s1 = starts[step], starts[step + 1]
    ks = [r for r in rows if s0 <= r[1] < s1]
    print(f"step wall {(s1 - s0) / 1e6:.3f} ms, {len(ks)} kernels")
    for sid in sorted({r[3] for r in ks}):
        sel = [r for r in ks if r[3] == sid]
        busy = sum(r[2] - r[1] for r in sel)
        print(f" stream {sid}: {len(sel)} kernels, busy {busy / 1e6:.3f} ms, span {(sel[0][1] - s0) / 1e6:.3f}..{(sel[-1][2] - s0) / 1e6:.3f} ms")
    main_s = [r for r in ks if r[3] == ks[0][3]]
    gaps = []
    for a, b in zip(main_s, main_s[1:]):
        g = b[1] - a[2]
        if g > 5000:
            gaps.append((g, (a[2] - s0) / 1e6, a[0].split("(")[0][:40], b[0].split("(")[0][:40]))
    tot = sum(max(0, b[1] - a[2]) for a, b in zip(main_s, main_s[1:]))
    print(f" main-stream idle between kernels: {tot / 1e6:.3f} ms; gaps > 5 us: {len(gaps)}")
    for g, t, a, b in sorted(gaps, reverse=True)[:25]:
        print(f"   {g / 1e3:8.1f} us at {t:7.3f} ms  after {a}  before {b}")
    # phases: forward ends at the first kernel whose name contains 'bwd' or 'recon_bwd'
    tb = next((r[1] for r in ks if "bwd" in r[0]), s1)
    to = next((r[1] for r in ks if "adamw" in r[0]), s1)
    print(f" forward {(tb - s0) / 1e6:.3f} ms | backward {(to - tb) / 1e6:.3f} ms | optimizer+tail {(s1 - to) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
