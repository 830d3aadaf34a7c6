#!/bin/bash
(cd build/r02 && python tools/gemm_bench.py --iters 30 2>/dev/null) > /tmp/gb_r02.txt
python tools/gemm_bench.py --iters 30 2>/dev/null > /tmp/gb_r03.txt
paste -d'|' /tmp/gb_r02.txt /tmp/gb_r03.txt | cut -c1-75,100-175
