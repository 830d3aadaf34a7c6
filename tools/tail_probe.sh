#!/bin/bash
# what sits between the last backward kernel and the AdamW launches: kernel + memory-copy + HIP API trace of a few bench steps
out=gpurun_out/tail_probe; mkdir -p $out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d /tmp/prof_tail -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1 > /dev/null 2> $GRAFT_REPO_ROOT/$out/err.txt )
db=$(find /tmp/prof_tail -name '*.db' | head -1)
python - $db > $out/report.txt 2>&1 <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if not t.startswith('rocpd_')][:40])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
ad = [r for r in rows if 'adamw' in r[0]]
big = [r for r in ad if r[2] - r[1] > 100000][-2]   # the large AdamW launch of the second-to-last step
t1 = big[1]; t0 = t1 - 400000
print("window", t0, t1)
for r in rows:
    if t0 <= r[1] <= t1 + 1000: print("K %9.1f %9.1f %s" % ((r[1]-t0)/1e3, (r[2]-t0)/1e3, r[0][:50]))
kc = [d[1] for d in c.execute("pragma table_info(kernels)")]
rc = [d[1] for d in c.execute("pragma table_info(regions)")]
print("kernels cols", kc); print("regions cols", rc)
# host lead: the k-th launch call of a thread is the k-th kernel of the trace that carries this corr/stack id; simplest robust match: by order
ks = [r for r in c.execute("select name, start, end, stack_id, corr_id from kernels where start between ? and ? order by start", (t0, t1 + 1000))]
print("sample kernel ids", [(k[0][:12], k[3], k[4]) for k in ks[-3:]])
for k in ks[-6:]:
    for key, val in (("stack_id", k[3]), ("corr_id", k[4])):
        if key in rc and val:
            rr = c.execute(f"select name, start, end, tid from regions where {key} = ? order by start", (val,)).fetchall()
            for r in rr[:3]:
                print("L %-26s gpu start %9.1f | %-8s api %-24s %9.1f .. %9.1f tid %s lead %.1f us" % (k[0][:26], (k[1]-t0)/1e3, key, str(r[0])[:24], (r[1]-t0)/1e3, (r[2]-t0)/1e3, r[3], (k[1]-r[2])/1e3))
PY
tail -40 $out/report.txt | cut -c1-230
