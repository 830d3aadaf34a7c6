tag=r02c; out=gpurun_out/$tag; mkdir -p $out
bash tools/pmc_round.sh $tag > /dev/null 2>&1
cp $out/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
cut -c1-200 $out/bench_default.json; head -3 $out/pmc_hbm_traffic.txt
