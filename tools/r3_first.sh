#!/bin/bash
# round 3, first GPU visit: the new LayerNorm-fold kernel tests, then the whole GPU suite, then the default bench line
out=gpurun_out/r3a; mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lnfold or resid_stats or fold_weights or row_statistics or 8bit or stack_boundaries" 2>&1 | tail -30 ) > $out/fold_tests.log
tail -5 $out/fold_tests.log
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $out/tests.log
tail -15 $out/tests.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cut -c1-600 $out/bench.json; tail -3 $out/bench.err
CSMAE_NO_LNFOLD=1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > $out/bench_nofold.json 2>> $out/bench.err
cut -c1-300 $out/bench_nofold.json
