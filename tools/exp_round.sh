#!/bin/bash
# One GPU-box visit for a GEMM experiment: GEMM parity tests, the micro-benchmark of this tree and of another checkout, interleaved step A/B.
#   tools/exp_round.sh TAG OTHER_TREE [rounds] [pytest -k expression]
tag=$1; other=$2; rounds=${3:-3}; kexpr=${4:-gemm}
out=gpurun_out/$tag; mkdir -p $out
( timeout 900 python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -5 ) > $out/tests.log
( timeout 300 python tools/gemm_bench.py 2>&1 ) > $out/gemm_this.txt
( cd $other && timeout 300 python tools/gemm_bench.py 2>&1 ) > $out/gemm_other.txt
( timeout 300 python tools/gemm_bench.py --cfg 2 --dbg 16 2>&1 | tail -1 ) > $out/loop_this.txt
( cd $other && timeout 300 python tools/gemm_bench.py --cfg 2 --dbg 16 2>&1 | tail -1 ) > $out/loop_other.txt
tools/ab_tree.sh $other $rounds > $out/ab.txt 2>&1
tail -3 $out/tests.log; paste $out/gemm_this.txt $out/gemm_other.txt | awk '{print $1,$2,$3,$(NF/2-3),$(NF/2-1),"|",$(NF-3),$(NF-1)}'; cat $out/loop_this.txt $out/loop_other.txt; cat $out/ab.txt
