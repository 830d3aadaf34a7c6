#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) into the per-kernel table kept under profiles/."""
import sqlite3
import sys


def main(db, out=None, skip_first_frac=0.0):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
    agg = {}
    for name, s, e, gx, wx, vg, lds in rows:
        short = name.split("(")[0]
        if short.startswith("void "):
            short = short[5:]
        d = agg.setdefault(short, dict(n=0, t=0, mn=1e30, mx=0, vg=vg, lds=lds))
        dt = e - s
        d["n"] += 1; d["t"] += dt; d["mn"] = min(d["mn"], dt); d["mx"] = max(d["mx"], dt)
    tot = sum(d["t"] for d in agg.values())
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db}", f"# total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches",
             f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'lds':>7s}"]
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
        lines.append(f"{k[:72]:72s} {d['n']:7d} {d['t'] / 1e6:10.3f} {d['t'] / d['n'] / 1e3:10.2f} {d['mn'] / 1e3:9.2f} {d['mx'] / 1e3:9.2f} {100 * d['t'] / tot:6.2f} {d['vg']:5d} {d['lds']:7d}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
