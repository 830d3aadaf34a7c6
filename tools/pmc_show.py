#!/usr/bin/env python3
"""Print per-kernel PMC counters (first dispatch of each kernel name) from one or more rocpd DBs written by rocprofv3 --pmc."""
import collections
import sqlite3
import sys

acc = collections.OrderedDict()
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, grid_size, counter_name, value, duration from counters_collection order by dispatch_id").fetchall()
    seen = {}
    for did, kn, gs, cn, v, dur in rows:
        key = kn.split("(")[0].replace("void ", "")[:50]
        if key not in seen:
            seen[key] = did
        if seen[key] != did:
            continue
        d = acc.setdefault(key, collections.OrderedDict())
        d.setdefault("dur_us", dur / 1e3)
        d[cn] = d.get(cn, 0) + v
for k, d in acc.items():
    print(k)
    wc = d.get("SQ_WAVE_CYCLES", 0)
    for cn, v in d.items():
        extra = f"  ({v / wc:.3f} of WAVE_CYCLES)" if wc and cn.startswith("SQ_") and cn not in ("SQ_WAVE_CYCLES",) and "INSTS" not in cn and "MFMA_BUSY" not in cn and "BUSY_CYCLES" not in cn else ""
        print(f"    {cn:34s} {v:16.4g}{extra}")
