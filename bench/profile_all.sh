#!/bin/bash
# The round's measurements of record, on the GPU box from the repo root (copy what is to be judged from
# gpurun_out/<tag>/ into profiles/<round>/):   bash bench/profile_all.sh <tag>
#   1. python bench.py                                         -> bench.json           (the default line, C3)
#   2. rocprofv3 --kernel-trace --stats -- python bench.py     -> stats_kernel_stats.csv, bench_under_rocprof.json
#   3. PMC passes of ONE full C3 frame (bench/pmc.sh) + TA/TD busy and stall counters (bench/pmc_stall.sh)
#   4. C4 (--scene tenm): bench line and PMC passes of one full frame
#   5. bench lines of C1, C2 and C5
tag=$1
out=gpurun_out/$tag
mkdir -p $out
# step <name> <seconds> <stdout file> <stderr file> command...
step() { name=$1; secs=$2; so=$3; se=$4; shift 4; echo "== $name"; timeout -k 10 $secs "$@" > $so 2> $se; rc=$?; echo "== $name rc=$rc"; case $rc in 124|137|134|139|143) exit $rc;; esac; }
step bench 400 $out/bench.json $out/bench.err python bench.py; tail -c 600 $out/bench.json
step stats 400 $out/bench_under_rocprof.json $out/bench_under_rocprof.err rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python bench.py
cut -c1-160 $out/stats_kernel_stats.csv | head -6
step pmc_c3 900 $out/pmc_c3.log $out/pmc_c3.err bash bench/pmc.sh ${tag}_c3 --steps 1 --warmup 0 --no-cpu-baseline --no-count; cp gpurun_out/pmc_${tag}_c3/summary.json $out/pmc_summary_sponza.json
step stall_c3 600 $out/pmc_ta_td_stall_sponza.txt $out/stall_c3.err bash bench/pmc_stall.sh ${tag}_c3s --steps 1 --warmup 0 --no-cpu-baseline --no-count
step bench_c4 600 $out/bench_tenm.json $out/bench_tenm.err python bench.py --scene tenm --steps 2 --warmup 1; tail -c 400 $out/bench_tenm.json
step pmc_c4 1200 $out/pmc_c4.log $out/pmc_c4.err bash bench/pmc.sh ${tag}_c4 --scene tenm --steps 1 --warmup 0 --no-cpu-baseline --no-count; cp gpurun_out/pmc_${tag}_c4/summary.json $out/pmc_summary_tenm.json
step stall_c4 900 $out/pmc_ta_td_stall_tenm.txt $out/stall_c4.err bash bench/pmc_stall.sh ${tag}_c4s --scene tenm --steps 1 --warmup 0 --no-cpu-baseline --no-count
step bench_c1 200 $out/bench_cornell.json $out/bench_cornell.err python bench.py --scene cornell --steps 20 --warmup 3
step bench_c2 200 $out/bench_bunny.json $out/bench_bunny.err python bench.py --scene bunny --steps 5 --warmup 1
step bench_c5 600 $out/bench_glossy.json $out/bench_glossy.err python bench.py --scene glossy --steps 1 --warmup 0; tail -c 300 $out/bench_glossy.json
echo "== done"
