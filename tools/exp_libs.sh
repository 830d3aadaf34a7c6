#!/bin/bash
# One GPU-box visit for variant libraries of the GEMM (tools/gemm_variants.sh): parity tests and micro-benchmark per library, phase clocks
# of the -DGEMM_TIMING builds, interleaved step A/B.   usage: tools/exp_libs.sh TAG "lib1.so lib2.so" ["timing names"] [rounds]
tag=$1; libs=$2; timing=$3; rounds=${4:-3}
out=gpurun_out/$tag; mkdir -p $out; rm -f $out/ab.txt
for l in $libs; do
  n=$(basename $l .so)
  ( CSMAE_LIB_PATH=$PWD/$l timeout 600 python -m pytest tests -m gpu -x -q -k gemm 2>&1 | tail -2 ) > $out/tests_$n.log
  ( CSMAE_LIB_PATH=$PWD/$l timeout 300 python tools/gemm_bench.py 2>&1 ) > $out/gemm_$n.txt
  echo "$n: $(tail -1 $out/tests_$n.log) | $(tail -1 $out/gemm_$n.txt)"
done
[ -n "$timing" ] && python tools/gemm_cycles.py -1 $timing 2>&1 | tee $out/cycles.txt
tools/ab_lib_env.sh $out/ab.txt "$libs" "X=1" $rounds > /dev/null 2>&1
cat $out/ab.txt
