#!/bin/bash
# HBM-traffic PMC passes of the default bench command (separate --pmc passes, --kernel-trace only: MI355X_MICROARCH.md).  usage: tools/pmc_round.sh TAG
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${tag}_$c -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 0 > /dev/null 2> $GRAFT_REPO_ROOT/$out/pmc_$c.err
done
cd $GRAFT_REPO_ROOT
f=$(find /tmp/pmc_${tag}_FETCH_SIZE -name '*.db' | head -1); w=$(find /tmp/pmc_${tag}_WRITE_SIZE -name '*.db' | head -1)
python tools/pmc_traffic.py $f $w 13 $out/pmc_hbm_traffic.txt $out/pmc_traffic.json | head -30
