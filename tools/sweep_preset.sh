#!/bin/bash
# step time of a bench preset under env-variable settings, interleaved on one box.  usage: tools/sweep_preset.sh OUT "PRESET ARGS" "VAR=a VAR=b ..." [rounds]
out=$1; pargs=$2; settings=$3; rounds=${4:-2}
run() { timeout 600 env $1 python bench.py $pargs --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$s $(run $s)" | tee -a $out; done
done
