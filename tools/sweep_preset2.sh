#!/bin/bash
# step time of a bench preset under env settings, interleaved.  usage: tools/sweep_preset2.sh OUT PRESET "A=1 B=2+C=3" [rounds] [extra bench args]
out=$1; preset=$2; settings=$3; rounds=${4:-2}; shift 4
run() { timeout 400 env ${1//+/ } python bench.py --preset $preset --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$preset $s $(run $s "$@")" | tee -a $out; done
done
