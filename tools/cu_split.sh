#!/bin/bash
# CU partition between the main chain and the weight-gradient stream: step time per split, interleaved on one box.
#   usage: tools/cu_split.sh OUT [rounds] ["setting setting ..."]     a setting = the value of CSMAE_DEBUG (comma-separated key=value pairs; "-" = none)
# (superseded by tools/cu_split2.sh, whose control is valid: this one runs the main chain on the legacy null stream — DESIGN §5 round 5)
out=$1; rounds=${2:-3}
settings=${3:-"- dw_cus=12 dw_cus=16 dw_cus=20 dw_cus=24 dw_cus=8,main_cus=64:256 dw_cus=12,main_cus=96:256 dw_cus=16,main_cus=128:256"}
run() { v=$1; [ "$v" = "-" ] && v=""; timeout 200 env CSMAE_DEBUG="$v" python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['loss'])"; }
for i in $(seq $rounds); do
  for s in $settings; do echo "$s $(run $s)" | tee -a $out; done
done
