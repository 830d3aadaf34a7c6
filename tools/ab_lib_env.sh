#!/bin/bash
# step time of library builds x env settings, interleaved on one box.  usage: tools/ab_lib_env.sh OUT "libA.so libB.so" "VAR=a VAR=b" [rounds]
out=$1; libs=$2; settings=$3; rounds=${4:-2}
run() { timeout 300 env CSMAE_LIB_PATH=$PWD/$1 $2 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  for l in $libs; do for s in $settings; do echo "$l $s $(run $l $s)" | tee -a $out; done; done
done
