#!/bin/bash
# A/B the step time of this tree against another checkout of the repository (its own Python AND its own library), interleaved on one GPU box:
#   git worktree add build/r02 <rev> && (cd build/r02 && make -j8)     # in the build container: build/ travels with gpurun
#   tools/ab_tree.sh build/r02 [rounds]
other=$1; rounds=${2:-3}
run() { (cd $1 && timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); }
for i in $(seq $rounds); do echo "this  $(run .)"; echo "other $(run $other)"; done
