#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "fp8" 2>&1 | tail -8 ) > $out/tests.log; tail -4 $out/tests.log
for v in A=1 CSMAE_FP8_TWO_STAGE=1 A=1 CSMAE_FP8_TWO_STAGE=1; do echo "$v $(env $v timeout 600 python bench.py --preset huge14 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['dominant_kernel']['by_layout'].get('gemm_fp8_NT'))")" | tee -a $out/ab_fp8.txt; done
