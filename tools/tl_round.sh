#!/bin/bash
# kernel trace of the bench step + per-stream timeline + per-kernel stats.  usage: tools/tl_round.sh TAG [env assignments...]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err )
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python tools/rocpd_stats.py $db $out/kernel_stats.txt > /dev/null
python tools/step_timeline.py $db > $out/timeline.txt 2>&1
python tools/step_dump.py $db $out/step_dump.txt > /dev/null 2>&1
head -6 $out/timeline.txt; tail -1 $out/timeline.txt
