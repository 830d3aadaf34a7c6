#!/bin/bash
# step time under env settings, interleaved on one box; a setting may hold several variables joined by '+'.  usage: tools/sweep_env2.sh OUT "A=1 A=1+B=2 ..." [rounds]
out=$1; settings=$2; rounds=${3:-2}
run() { timeout 300 env ${1//+/ } python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$s $(run $s)" | tee -a $out; done
done
