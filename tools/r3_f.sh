#!/bin/bash
out=gpurun_out/r3f; mkdir -p $out
( timeout 900 python -m pytest tests/test_train_gpu.py -q -x -k "bench_two_ranks or forced or two_rank" 2>&1 | tail -5 ) > $out/tests.log; tail -3 $out/tests.log
bash tools/ab_env.sh CSMAE_ZERO_MAIN=1 3 | tee $out/ab_zero.txt
bash tools/roofline_round.sh r3f | tee $out/roofline_head.txt
