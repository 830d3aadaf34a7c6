#!/bin/bash
out=gpurun_out/r3b; mkdir -p $out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lnfold or resid_stats or fold_weights or row_statistics" 2>&1 | tail -30 ) > $out/fold_tests.log
tail -5 $out/fold_tests.log
bash tools/prof_env.sh r3b_fold
bash tools/prof_env.sh r3b_nofold CSMAE_NO_LNFOLD=1
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('fold', json.loads(sys.stdin.read())['ms_per_step'])"
  CSMAE_NO_LNFOLD=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('nofold', json.loads(sys.stdin.read())['ms_per_step'])"
done
