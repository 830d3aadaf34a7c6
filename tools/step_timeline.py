#!/usr/bin/env python3
"""Per-stream timeline of one training step from a rocprofv3 rocpd DB: busy time per stream, idle gaps on the main stream,
and a phase table.  With a DB taken with `--kernel-trace --marker-trace --hip-runtime-trace` the phases come from the roctx ranges the product
emits (csmae_hip/trace.py: csmae.forward / csmae.backward.junction / .decoder / .encoder / csmae.optimizer / csmae.exchange): a kernel belongs to
the innermost range its launch call was made in; without markers they are guessed from kernel names.  usage: step_timeline.py results.db [step_index]"""
import sqlite3
import sys


def main(db, step=-2, phases_only=False):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    starts = [r[1] for r in rows if "crop_resize" in r[0]]
    s0, s1 = starts[step], starts[step + 1]
    ks = [r for r in rows if s0 <= r[1] < s1]
    if phases_only:   # (a DB taken with --hip-runtime-trace: the host falls behind under API tracing, so its gaps are the profiler's — only the attribution is used)
        if not marker_phases(c, ks, s0):
            cats = c.execute("select category, count(*) from regions group by category").fetchall()
            print(f" no csmae.* roctx ranges in this DB (region categories: {cats})")
        return
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
    if marker_phases(c, ks, s0):
        return
    # phases: forward ends at the first kernel whose name contains 'bwd' or 'recon_bwd'
    tb = next((r[1] for r in ks if "bwd" in r[0]), s1)
    to = next((r[1] for r in ks if "adamw" in r[0]), s1)
    print(f" forward {(tb - s0) / 1e6:.3f} ms | backward {(to - tb) / 1e6:.3f} ms | optimizer+tail {(s1 - to) / 1e6:.3f} ms")


def marker_phases(c, ks, s0):
    """Phase table from roctx ranges; False when the DB holds none."""
    try:
        rng = c.execute("select name, start, end, id from regions where name like 'csmae.%' order by start").fetchall()
        if not rng:   # rocprofv3 (ROCm 7): the region is named after the API call (roctxThreadRangeA), the message sits in extdata: {"message": "csmae.forward"}
            import json
            for ext, st, en, rid in c.execute("select extdata, start, end, id from regions where category like 'MARKER%' order by start").fetchall():
                try:
                    msg = json.loads(ext).get("message", "")
                except (ValueError, TypeError):
                    msg = ""
                if msg.startswith("csmae."):
                    rng.append((msg, st, en, rid))
        if not rng:
            return False
        ids = {r[0] for r in c.execute("select stack_id from kernels where start >= ? and start < ?", (ks[0][1], ks[-1][1] + 1))}
        launches = dict(c.execute("select stack_id, start from regions where category like 'HIP%' and name like '%Launch%'").fetchall())
        kmeta = c.execute("select name, start, end, stream_id, stack_id from kernels where start >= ? and start <= ? order by start", (ks[0][1], ks[-1][1])).fetchall()
    except sqlite3.Error as e:
        print(f" (marker phases unavailable: {e})")
        return False
    import bisect
    starts = [r[1] for r in rng]
    phases = {}
    for name, st, en, sid, stack in kmeta:
        t = launches.get(stack)
        if t is None:
            continue
        # innermost range containing the launch call: the latest-starting range with start <= t < end
        i = bisect.bisect_right(starts, t) - 1
        while i >= 0 and not (rng[i][1] <= t < rng[i][2]):
            i -= 1
        ph = rng[i][0] if i >= 0 else "(outside any csmae range)"
        d = phases.setdefault(ph, dict(n=0, busy=0, first=st, last=en))
        d["n"] += 1; d["busy"] += en - st; d["first"] = min(d["first"], st); d["last"] = max(d["last"], en)
    print(" phases from roctx ranges (kernels by the range their launch call was made in; GPU span = first kernel start .. last kernel end):")
    for ph, d in sorted(phases.items(), key=lambda kv: kv[1]["first"]):
        print(f"   {ph:34s} {d['n']:4d} kernels  GPU span {(d['first'] - s0) / 1e6:7.3f} .. {(d['last'] - s0) / 1e6:7.3f} ms ({(d['last'] - d['first']) / 1e6:6.3f} ms)  kernel time {d['busy'] / 1e6:7.3f} ms")
    return True


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--phases-only"]
    main(args[0], int(args[1]) if len(args) > 1 else -2, phases_only="--phases-only" in sys.argv)
