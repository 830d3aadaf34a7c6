#!/bin/bash
# Everything the judged profiles/ files come from, on one box: default bench line, kernel stats of the same command (overlapped and
# serialised), HBM-traffic PMC passes, preset lines.  usage: tools/profile_round.sh TAG
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
# a profile names the kernel sources it measured: refuse to run when the library was not built from the sources in the tree
python -c "
import sys
sys.path.insert(0, 'cross-scale-mae_amd'); sys.path.insert(0, 'tools')
import csmae_hip
from csrc_hash import csrc_hash
lib, src = csmae_hip.source_hash(), csrc_hash()
sys.exit(0 if lib == src else f'libcsmae_hip.so was built from csrc {lib[:16]}, the tree holds {src[:16]}: rebuild (make) before profiling')
" || exit 1
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find /tmp/prof_$tag -name '*.db' | head -1) $out/step_kernel_stats.txt > /dev/null
python tools/step_timeline.py $(find /tmp/prof_$tag -name '*.db' | head -1) > $out/step_timeline.txt 2>&1
# ... and the phases from the roctx ranges the product emits (csmae_hip/trace.py): kernel + marker + HIP-API trace of a short run (attribution only:
# under API tracing the host falls behind the GPU, so that run's gaps are the profiler's)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace -d /tmp/profm_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2 > /dev/null 2>&1 )
python tools/step_timeline.py $(find /tmp/profm_$tag -name '*.db' | head -1) --phases-only >> $out/step_timeline.txt 2>&1
( cd /tmp && CSMAE_DW_MAIN=1 CSMAE_FWD_ONE_STREAM=1 CSMAE_OPT_MAIN=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/profs_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find /tmp/profs_$tag -name '*.db' | head -1) $out/step_serialised_kernel_stats.txt > /dev/null
bash tools/roofline_round.sh $tag > /dev/null 2>&1   # per-kernel HBM GB/s + MFMA utilisation (kernel_roofline.txt) and the HBM traffic table / pmc_traffic.json
for p in large large4; do timeout 600 python bench.py --preset $p --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $out/bench_presets.txt; done
for d in bf16 fp8; do timeout 600 python bench.py --preset huge14 --dtype $d --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> $out/bench_presets.txt; done
cp $out/pmc_traffic.json profiles/pmc_traffic.json   # (on the GPU box only: the bench line below then carries roofline.traffic for these sources; copy it into profiles/ by hand afterwards)
timeout 600 python bench.py 2>/dev/null | tail -1 > $out/bench_final.json
# per-kernel table of the fp8 step (BASELINE.json configs[4]: ViT-H/14, 256 per GPU, fp8 MFMA GEMMs), taken on the same sources
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/proff8_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --preset huge14 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find /tmp/proff8_$tag -name '*.db' | head -1) $out/huge14_fp8_kernel_stats.txt > /dev/null
[ -n "$GRAPH_PROBE" ] && timeout 400 python tools/graph_probe.py 2>&1 | grep -E "^eager|^graph|^captured|^loss" > $out/graph_vs_eager.txt
cut -c1-300 $out/bench_default.json; head -4 $out/pmc_hbm_traffic.txt
