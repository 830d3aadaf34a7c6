#!/bin/bash
# step time under env-variable settings, interleaved on one box.  usage: tools/sweep_env.sh OUT "VAR=a VAR=b ..." [rounds]
out=$1; settings=$2; rounds=${3:-2}
run() { timeout 300 env $1 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$s $(run $s)" | tee -a $out; done
done
