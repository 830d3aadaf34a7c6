#!/bin/bash
# shader clock / power while the bench step runs (is the step power-limited?).  usage: tools/clk_watch.sh [bench args]
( timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 3000 --warmup 5 "$@" 2>/dev/null | tail -1 | cut -c1-200 ) &
bp=$!
sleep 40
for i in $(seq 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done
wait $bp
