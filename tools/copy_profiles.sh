#!/bin/bash
# copy a tools/profile_round.sh result set from gpurun_out/TAG into profiles/ under the round's names.  usage: tools/copy_profiles.sh TAG rNN
tag=$1; r=$2; src=gpurun_out/$tag
for f in bench_default.json bench_final.json bench_presets.txt graph_vs_eager.txt kernel_roofline.txt pmc_hbm_traffic.txt step_kernel_stats.txt step_serialised_kernel_stats.txt step_timeline.txt huge14_fp8_kernel_stats.txt; do
  [ -s $src/$f ] && cp $src/$f profiles/${r}_$f
done
[ -s $src/pmc_traffic.json ] && cp $src/pmc_traffic.json profiles/pmc_traffic.json
ls -la profiles/${r}_* profiles/pmc_traffic.json
