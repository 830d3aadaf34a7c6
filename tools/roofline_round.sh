#!/bin/bash
# profiles/rNN_kernel_roofline.txt: per-kernel HBM GB/s and MFMA utilisation of the bench step as numbers.  usage: tools/roofline_round.sh TAG
# Refuses to run when the loaded library was not built from the sources in the tree (a profile names the sources it measured).
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python - <<'PY' || exit 1
import sys
sys.path.insert(0, "cross-scale-mae_amd"); sys.path.insert(0, "tools")
import csmae_hip
from csrc_hash import csrc_hash
lib, src = csmae_hip.source_hash(), csrc_hash()
if lib != src:
    sys.exit(f"libcsmae_hip.so was built from csrc {lib[:16]}, the tree holds {src[:16]}: rebuild (make) before profiling")
PY
cmd="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/rl_${tag}_trace -o r -- $cmd > /dev/null 2> $GRAFT_REPO_ROOT/$out/rl_trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/rl_${tag}_$c -o r -- $cmd > /dev/null 2> $GRAFT_REPO_ROOT/$out/rl_$c.err
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/rl_${tag}_sq -o r -- $cmd > /dev/null 2> $GRAFT_REPO_ROOT/$out/rl_sq.err
cd $GRAFT_REPO_ROOT
db() { find /tmp/rl_${tag}_$1 -name '*.db' | head -1; }
python tools/kernel_roofline.py $(db trace) $(db FETCH_SIZE) $(db WRITE_SIZE) $(db sq) 13 $out/kernel_roofline.txt | head -45
python tools/pmc_traffic.py $(db FETCH_SIZE) $(db WRITE_SIZE) 13 $out/pmc_hbm_traffic.txt $out/pmc_traffic.json > /dev/null
