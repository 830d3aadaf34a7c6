#!/bin/bash
# kernel stats of a preset's step.  usage: tools/tl_preset.sh TAG <bench args...>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err )
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python tools/rocpd_stats.py $db $out/kernel_stats.txt > /dev/null
head -32 $out/kernel_stats.txt | cut -c1-150
