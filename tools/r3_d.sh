#!/bin/bash
out=gpurun_out/r3d; mkdir -p $out
python tools/lnfold_bench.py | tee $out/lnfold_bench.txt
( timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "lnfold or fullsize or n128 or micro_bf16" 2>&1 | tail -8 ) > $out/tests.log
tail -4 $out/tests.log
( CSMAE_LNFOLD_Y=keep timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "fullsize or n128 or micro_bf16" 2>&1 | tail -8 ) > $out/tests_keep.log
tail -4 $out/tests_keep.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do
  echo "recompute $(run)"
  echo "keep $(CSMAE_LNFOLD_Y=keep run)"
  echo "nofold $(CSMAE_NO_LNFOLD=1 run)"
done
