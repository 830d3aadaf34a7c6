#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "gelu or epilogue" 2>&1 | tail -5 ) > $out/tests.log; tail -3 $out/tests.log
python tools/lnfold_bench.py 2>/dev/null | grep fc1
CSMAE_LIB_PATH=$PWD/build/lib_base3.so python tools/lnfold_bench.py 2>/dev/null | grep fc1
bash tools/ab_lib_env.sh $out/ab.txt "build/lib_base3.so cross-scale-mae_amd/csmae_hip/libcsmae_hip.so" "A=1" 3
