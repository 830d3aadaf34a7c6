#!/usr/bin/env python3
"""Every kernel of one training step from a rocprofv3 rocpd DB, in start order: stream, start / end relative to the step (us), grid
size in workgroups, name.  usage: step_dump.py results.db out.txt [step_index]"""
import sqlite3
import sys


def main(db, out, step=-2):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
    starts = [r[1] for r in rows if "crop_resize" in r[0]]
    s0, s1 = starts[step], starts[step + 1]
    ks = [r for r in rows if s0 <= r[1] < s1]
    sids = {sid: i for i, sid in enumerate(sorted({r[3] for r in ks}, key=lambda s: min(r[1] for r in ks if r[3] == s)))}
    with open(out, "w") as f:
        f.write(f"# step wall {(s1 - s0) / 1e3:.1f} us, {len(ks)} kernels; columns: stream start_us end_us dur_us workgroups name\n")
        for name, s, e, sid, gx, wx in ks:
            short = name.split("(")[0].replace("void ", "")[:60]
            f.write(f"{sids[sid]} {(s - s0) / 1e3:9.1f} {(e - s0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gx // max(wx, 1):6d} {short}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -2)
