#!/bin/bash
# One GPU-box visit: tests, default bench line, kernel trace of the same command, per-stream timeline.  usage: tools/gpu_round.sh TAG [pytest-args]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15 ) > $out/tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$out/prof.err )
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python tools/rocpd_stats.py $db $out/kernel_stats.txt > /dev/null
python tools/step_timeline.py $db > $out/timeline.txt 2>&1
python tools/step_dump.py $db $out/step_dump.txt > /dev/null 2>&1
tail -3 $out/tests.log; cat $out/bench.json | cut -c1-400
