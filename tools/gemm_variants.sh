#!/bin/bash
# Build variants of the GEMM library (compile-time -D flags of csrc/gemm.hip: -DGEMM_TIMING, -DGEMM_ABL=..) next to the product build.
# usage: tools/gemm_variants.sh NAME1="-DPP_ABL=1" NAME2="-DPP_VAR=2" ...  ->  build/abl/libcsmae_pp_NAME.so
set -e
mkdir -p build/abl
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c cross-scale-mae_amd/csrc/gemm.hip -o build/abl/gemm_pp_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%=*}
  hipcc --offload-arch=gfx950 -shared -fPIC build/obj/api.o build/obj/attention.o build/obj/fp8.o build/abl/gemm_pp_$name.o build/obj/loss.o build/obj/norm.o build/obj/optim.o build/obj/tokens.o -o build/abl/libcsmae_pp_$name.so
done
ls build/abl/libcsmae_pp_*.so
