#!/bin/bash
# A/B the step time of the current build with and without an environment switch, interleaved: tools/ab_env.sh CSMAE_GEMM_MIXED=0 [rounds]
sw=$1; rounds=${2:-3}
run() { timeout 300 env "$@" python bench.py --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $rounds); do
  echo "with $sw: $(run $sw)"
  echo "default: $(run A=1)"
done
