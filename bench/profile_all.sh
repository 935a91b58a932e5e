#!/bin/bash
# The round's measurements of record, on the GPU box from the repo root (copy what is to be judged from
# gpurun_out/<tag>/ into profiles/<round>/ and gpurun_out/<tag>/pmc_traffic.json to profiles/pmc_traffic.json):
#   bash bench/profile_all.sh <tag>
#   1. python bench.py                                         -> bench_before_pmc.json (its in-kernel lane-load count feeds step 3)
#   2. PMC passes of ONE full C3 frame (bench/pmc.sh) + TA/TD busy and stall counters (bench/pmc_stall.sh)
#   3. profiles/pmc_traffic.json updated from them (on the box; copied to <out>/pmc_traffic.json)
#   4. python bench.py                                         -> bench.json           (the default line, C3, PMC fields at this kernel hash)
#   5. rocprofv3 --kernel-trace --stats -- python bench.py     -> stats_kernel_stats.csv, bench_under_rocprof.json
#   6. C4 (--scene tenm): PMC passes of one full frame, then its bench line
#   7. bench lines of C1, C2 and C5; C5 progressive (reduce every 16 spp) with a one-rank RCCL exchange, both exchange paths;
#      C3 with the one-rank tile exchange through RCCL
tag=$1
out=gpurun_out/$tag
mkdir -p $out
P="--steps 1 --warmup 0 --no-cpu-baseline --no-count --no-iter"
# step <name> <seconds> <stdout file> <stderr file> command...
step() { name=$1; secs=$2; so=$3; se=$4; shift 4; echo "== $name"; timeout -k 10 $secs "$@" > $so 2> $se; rc=$?; echo "== $name rc=$rc"; case $rc in 124|137|134|139|143) exit $rc;; esac; }
step bench_pre 400 $out/bench_before_pmc.json $out/bench_before_pmc.err python bench.py --no-cpu-baseline
step pmc_c3 900 $out/pmc_c3.log $out/pmc_c3.err bash bench/pmc.sh ${tag}_c3 $P; cp gpurun_out/pmc_${tag}_c3/summary.json $out/pmc_summary_sponza_full_frame.json
step stall_c3 600 $out/pmc_ta_td_stall_sponza.txt $out/stall_c3.err bash bench/pmc_stall.sh ${tag}_c3s $P
python bench/pmc_traffic_update.py $out/pmc_summary_sponza_full_frame.json sponza "round 3 final source state, full 1080p 256-spp frame" > /dev/null &&
python bench/pmc_stall_update.py $out/pmc_ta_td_stall_sponza.txt sponza $out/bench_before_pmc.json > /dev/null
step pmc_c4 1200 $out/pmc_c4.log $out/pmc_c4.err bash bench/pmc.sh ${tag}_c4 --scene tenm $P; cp gpurun_out/pmc_${tag}_c4/summary.json $out/pmc_summary_tenm_full_frame.json
step stall_c4 900 $out/pmc_ta_td_stall_tenm.txt $out/stall_c4.err bash bench/pmc_stall.sh ${tag}_c4s --scene tenm $P
python bench/pmc_traffic_update.py $out/pmc_summary_tenm_full_frame.json tenm "round 3 final source state, full 4K 64-spp frame" > /dev/null &&
python bench/pmc_stall_update.py $out/pmc_ta_td_stall_tenm.txt tenm > /dev/null
cp profiles/pmc_traffic.json $out/pmc_traffic.json
step bench 400 $out/bench.json $out/bench.err python bench.py; tail -c 600 $out/bench.json
step stats 400 $out/bench_under_rocprof.json $out/bench_under_rocprof.err rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python bench.py
cut -c1-160 $out/stats_kernel_stats.csv | head -8
step bench_c4 600 $out/bench_tenm.json $out/bench_tenm.err python bench.py --scene tenm --steps 2 --warmup 1; tail -c 400 $out/bench_tenm.json
step bench_c1 200 $out/bench_cornell.json $out/bench_cornell.err python bench.py --scene cornell --steps 20 --warmup 3
step bench_c2 200 $out/bench_bunny.json $out/bench_bunny.err python bench.py --scene bunny --steps 5 --warmup 1
step bench_c5 600 $out/bench_glossy.json $out/bench_glossy.err python bench.py --scene glossy --steps 1 --warmup 0; tail -c 300 $out/bench_glossy.json
X="--steps 1 --warmup 0 --no-cpu-baseline --no-count --no-iter --force-exchange"
step c5_prog_native 600 $out/bench_glossy_progressive_native.json $out/c5pn.err python bench.py --scene glossy --shard samples --reduce-every 16 --exchange native $X
step c5_prog_torch 600 $out/bench_glossy_progressive_torch.json $out/c5pt.err python bench.py --scene glossy --shard samples --reduce-every 16 --exchange torch $X
step c5_prog_tiles 600 $out/bench_glossy_progressive_tiles_native.json $out/c5ptl.err python bench.py --scene glossy --shard tiles --reduce-every 16 --exchange native $X
step c3_exch_torch 300 $out/bench_sponza_one_rank_rccl_torch.json $out/c3xt.err python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count --no-iter --force-exchange --exchange torch
step c3_exch_native 300 $out/bench_sponza_one_rank_rccl_native.json $out/c3xn.err python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count --no-iter --force-exchange --exchange native
echo "== done"
