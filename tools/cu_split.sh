#!/bin/bash
# CU partition between the main chain and the weight-gradient stream: step time per split, interleaved on one box.
#   usage: tools/cu_split.sh OUT [rounds] ["setting setting ..."]     a setting = comma-separated VAR=value pairs
out=$1; rounds=${2:-3}
settings=${3:-"A=1 CSMAE_DW_CUS=12 CSMAE_DW_CUS=16 CSMAE_DW_CUS=20 CSMAE_DW_CUS=24 CSMAE_DW_CUS=8,CSMAE_MAIN_CUS=64:256 CSMAE_DW_CUS=12,CSMAE_MAIN_CUS=96:256 CSMAE_DW_CUS=16,CSMAE_MAIN_CUS=128:256"}
run() { timeout 200 env $(echo $1 | tr ',' ' ') python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$s $(run $s)" | tee -a $out; done
done
