#!/bin/bash
# SQ counters of the GEMM family on the step's shapes (tools/pmc_sq2.txt: two --pmc passes, --kernel-trace only).  usage: tools/pmc_sq_round.sh TAG
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_sq2.txt --kernel-trace -d /tmp/sq_$tag -o r -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 2 > /dev/null 2> $GRAFT_REPO_ROOT/$out/sq.err )
{ echo "# rocprofv3 -i tools/pmc_sq2.txt --kernel-trace -- python tools/gemm_bench.py --iters 2   (first dispatch of each kernel name; counters slow the dispatch)"; python tools/pmc_show.py $(find /tmp/sq_$tag -name '*.db'); } > $out/pmc_sq_gemm.txt
head -40 $out/pmc_sq_gemm.txt
