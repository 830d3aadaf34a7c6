#!/bin/bash
out=gpurun_out/r3e; mkdir -p $out
( timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > $out/tests.log
tail -6 $out/tests.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do echo "new $(run)"; done
bash tools/prof_env.sh r3e_prof
