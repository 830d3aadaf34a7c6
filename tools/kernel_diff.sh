#!/bin/bash
# Per-kernel time of the bench step in this tree against another checkout (tools/ab_tree.sh), both under rocprofv3 --kernel-trace on one box:
# the kernels whose time per step moved most.  (Round 3 found a LayerNorm backward that had lost an occupancy step this way.)   usage: tools/kernel_diff.sh build/r02
other=$1
export TMPDIR=/tmp
for t in other this; do
  root=$GRAFT_REPO_ROOT; [ $t = other ] && root=$GRAFT_REPO_ROOT/$other
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cmp_$t -o r -- python $root/bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 > /dev/null 2>&1 )
  python tools/rocpd_stats.py $(find /tmp/cmp_$t -name '*.db' | head -1) gpurun_out/cmp_$t.txt > /dev/null
done
python - <<'PY'
import re
def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r'(.{74})\s+(\d+)\s+([\d.]+)\s+([\d.]+)', ln)
        if m and not ln.startswith(('#', 'kernel ')):
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load('gpurun_out/cmp_other.txt'), load('gpurun_out/cmp_this.txt')
steps = 22.0   # 10 settle + 2 warm-up + 10 timed
rows = [((b.get(k, (0, 0.0))[1] - a.get(k, (0, 0.0))[1]) / steps * 1e3, k) for k in set(a) | set(b)]
for d, k in sorted(rows, key=lambda r: -abs(r[0]))[:24]:
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    print(f"{d:+8.1f} us/step  {k[:62]:62s} other {ca:5d} x {ta:8.2f} ms | this {cb:5d} x {tb:8.2f} ms")
print(f"kernel time per step: other {sum(v[1] for v in a.values()) / steps:.2f} ms, this {sum(v[1] for v in b.values()) / steps:.2f} ms")
PY
