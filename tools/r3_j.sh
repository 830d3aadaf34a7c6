#!/bin/bash
run() { (cd $1 && timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for i in 1 2 3; do echo "r03 $(run .)"; echo "r02 $(run build/r02)"; done
echo "r03 zero-main $(CSMAE_ZERO_MAIN=1 run .)"
