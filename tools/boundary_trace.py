#!/usr/bin/env python3
"""Runs on the GPU box: what happens on the host and the device between the last backward kernel and the optimizer kernel."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
print("views:", views)
def cols(t):
    return [d[1] for d in c.execute(f"pragma table_info({t})")]
for v in ("regions", "memory_copies", "kernels"):
    if v in views:
        print(v, cols(v))
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
ad = [r for r in rows if "adamw" in r[0] and r[2] - r[1] > 100000]
t_ad = ad[-2][1]
prev = [r for r in rows if r[2] < t_ad][-3:]
t0 = prev[0][1]
print("window", (t_ad - t0) / 1e3, "us")
for r in prev:
    print(f"  kernel {(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:7.1f} us s{r[3]} {r[0][:50]}")
if "memory_copies" in views:
    for r in c.execute("select * from memory_copies where start between ? and ? order by start", (t0, t_ad + 1000000)).fetchall()[:20]:
        print("  memcpy", r)
if "regions" in views:
    cl = cols("regions")
    q = "select name, start, end, tid from regions where end >= ? and start <= ? order by start"
    regs = c.execute(q, (t0 - 2000000, t_ad + 100000)).fetchall()
    print(len(regs), "api calls in window (showing long ones and the last 60)")
    for r in regs:
        if r[2] - r[1] > 50000:
            print(f"  LONG {(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:8.1f} us tid {r[3]} {r[0]}")
    for r in regs[-60:]:
        print(f"  api {(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:8.1f} us tid {r[3]} {r[0]}")
