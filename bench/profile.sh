#!/bin/bash
# Produces the round's profile artefacts under gpurun_out/profile_<tag>/ (copy the summaries to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (python bench.py)
#   2. PMC passes (bench/pmc.sh) of ONE full-size launch (256 spp) of render_kernel
# Usage on the GPU box, from the repo root:  bash bench/profile.sh <tag>
tag=$1
out=gpurun_out/profile_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python bench.py > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err || { echo "stats run failed"; tail -3 $out/bench_under_rocprof.err; exit 1; }
cat $out/stats_kernel_stats.csv
bash bench/pmc.sh $tag --steps 1 --warmup 0 --no-cpu-baseline --no-count
cp gpurun_out/pmc_$tag/summary.json $out/pmc_summary.json
