#!/bin/bash
# Variant libraries of the two-workgroups-per-CU GEMM (csrc/gemm_k2.hip, -DK2_ABL=.. / other -D flags) on top of a -DGEMM_TIMING build of gemm.hip.
# usage: tools/k2_variants.sh NAME1="-DK2_ABL=1" ...  ->  build/abl/libcsmae_k2_NAME.so   (every variant carries -DGEMM_TIMING)
set -e
mkdir -p build/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGEMM_TIMING"
[ build/abl/gemm_timing.o -nt cross-scale-mae_amd/csrc/gemm.hip ] && [ build/abl/gemm_timing.o -nt cross-scale-mae_amd/csrc/gemm_common.h ] || hipcc $F -c cross-scale-mae_amd/csrc/gemm.hip -o build/abl/gemm_timing.o &
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  hipcc $F $flags -c cross-scale-mae_amd/csrc/gemm_k2.hip -o build/abl/gemm_k2_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  hipcc --offload-arch=gfx950 -shared -fPIC build/obj/api.o build/obj/attention.o build/obj/fp8.o build/abl/gemm_timing.o build/abl/gemm_k2_$name.o build/obj/loss.o build/obj/norm.o build/obj/optim.o build/obj/tokens.o -o build/abl/libcsmae_k2_$name.so
done
ls build/abl/libcsmae_k2_*.so
