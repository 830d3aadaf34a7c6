#!/bin/bash
export TMPDIR=/tmp
for t in r02 r03; do
  root=$GRAFT_REPO_ROOT; [ $t = r02 ] && root=$GRAFT_REPO_ROOT/build/r02
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cmp_$t -o r -- python $root/bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 > /dev/null 2>&1 )
  python tools/rocpd_stats.py $(find /tmp/cmp_$t -name '*.db' | head -1) gpurun_out/cmp_$t.txt > /dev/null
done
python - <<'PY'
import re
def load(p):
    d={}
    for ln in open(p):
        if ln.startswith('#') or ln.startswith('kernel'): continue
        m=re.match(r'(.{74})\s+(\d+)\s+([\d.]+)\s+([\d.]+)',ln)
        if m: d[m.group(1).strip()]=(int(m.group(2)),float(m.group(3)))
    return d
a,b=load('gpurun_out/cmp_r02.txt'),load('gpurun_out/cmp_r03.txt')
steps=22.0
rows=[]
for k in set(a)|set(b):
    ca,ta=a.get(k,(0,0.0)); cb,tb=b.get(k,(0,0.0))
    rows.append(((tb-ta)/steps*1e3,k,ca,ta,cb,tb))
for d,k,ca,ta,cb,tb in sorted(rows,key=lambda r:-abs(r[0]))[:22]:
    print(f"{d:+8.1f} us/step  {k[:60]:60s} r02 {ca:5d} {ta:8.2f} ms | r03 {cb:5d} {tb:8.2f} ms")
print("total", sum(v[1] for v in a.values())/steps, sum(v[1] for v in b.values())/steps)
PY
