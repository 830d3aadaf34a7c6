#!/usr/bin/env python3
"""Kernels of every stream in a time window of one step, from a rocprofv3 rocpd DB (start / end relative to the step's crop kernel).
usage: trace_window.py results.db T0_MS T1_MS [step_index]   (negative T0: the end of the step before)"""
import sqlite3
import sys


def main(db, t0, t1, step=-3):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    starts = [r[1] for r in rows if "crop_resize" in r[0]]
    s0 = starts[step]
    for name, st, en, sid in rows:
        if s0 + t0 * 1e6 <= st < s0 + t1 * 1e6:
            print(f"{(st - s0) / 1e3:10.1f} .. {(en - s0) / 1e3:10.1f} us  ({(en - st) / 1e3:7.1f})  s{sid}  {name.split('(')[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else -3)
