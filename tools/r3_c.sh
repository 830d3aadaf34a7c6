#!/bin/bash
out=gpurun_out/r3c; mkdir -p $out
( timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lnfold or resid_stats or fold_weights or row_statistics" 2>&1 | tail -30 ) > $out/fold_tests.log
tail -3 $out/fold_tests.log
python tools/lnfold_bench.py | tee $out/lnfold_bench.txt
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('fold', json.loads(sys.stdin.read())['ms_per_step'])"
  CSMAE_NO_LNFOLD=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print('nofold', json.loads(sys.stdin.read())['ms_per_step'])"
done
