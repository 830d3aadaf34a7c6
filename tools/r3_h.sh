#!/bin/bash
out=gpurun_out/r3h; mkdir -p $out
( timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "fp8" 2>&1 | tail -5 ) > $out/tests.log; tail -3 $out/tests.log
for d in bf16 fp8; do timeout 600 python bench.py --preset huge14 --dtype $d --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 | tee -a $out/bench_huge14.txt; done
