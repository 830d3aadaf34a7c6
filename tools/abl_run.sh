#!/bin/bash
# TN / NT main-loop ablations on the GPU box (see tools/gemm_ablate.sh): full-chip launches, main loop only (--dbg 16)
out=gpurun_out/$1; mkdir -p $out
export CSMAE_DEBUG=dw_slots=256
for b in 0 1 2 4 8; do
  echo "== abl $b dW main-loop-only" >> $out/abl.txt
  CSMAE_LIB_PATH=$PWD/build/abl/libcsmae_abl$b.so python tools/gemm_bench.py --only dW --cfg 4 --dbg 16 2>&1 | grep -v "^$" >> $out/abl.txt
done
for b in 0 1 2 4; do
  echo "== abl $b fwd(NT)+dX(NN) main-loop-only" >> $out/abl.txt
  CSMAE_LIB_PATH=$PWD/build/abl/libcsmae_abl$b.so python tools/gemm_bench.py --only dec --cfg 4 --dbg 16 2>&1 | grep -v "dW" >> $out/abl.txt
done
echo "== full kernels, product lib, 256 slots" >> $out/abl.txt
python tools/gemm_bench.py 2>&1 >> $out/abl.txt
export CSMAE_DEBUG=dw_slots=160
echo "== full kernels, product lib, 160 slots (dW only)" >> $out/abl.txt
python tools/gemm_bench.py --only dW 2>&1 >> $out/abl.txt
cat $out/abl.txt
