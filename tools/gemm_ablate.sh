#!/bin/bash
# Build main-loop ablation variants of the GEMM library (compile-time GEMM_ABL bits, csrc/gemm.hip) next to the product build.
# usage: tools/gemm_ablate.sh   ->  build/abl/libcsmae_abl<bits>.so for bits in 0 1 2 4 8
set -e
mkdir -p build/abl
for b in 0 1 2 4 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DGEMM_ABL=$b -c cross-scale-mae_amd/csrc/gemm.hip -o build/abl/gemm$b.o &
done
wait
for b in 0 1 2 4 8; do
  hipcc --offload-arch=gfx950 -shared -fPIC build/obj/api.o build/obj/attention.o build/abl/gemm$b.o build/obj/loss.o build/obj/norm.o build/obj/optim.o build/obj/tokens.o -o build/abl/libcsmae_abl$b.so
done
ls -la build/abl/*.so
