#!/bin/bash
# PMC passes of one full frame of C1, C2 and C5 (bench/pmc.sh + bench/pmc_stall.sh), folded into profiles/pmc_traffic.json on the box and
# copied to gpurun_out/<tag>/ with the bench lines that then carry them.   bash bench/pmc_other.sh <tag>
tag=$1
out=gpurun_out/$tag
mkdir -p $out
P="--steps 1 --warmup 0 --no-cpu-baseline --no-count --no-iter"
for scene in cornell bunny glossy; do
  timeout -k 10 300 python bench.py --scene $scene --steps 1 --warmup 0 --no-cpu-baseline --no-iter > $out/pre_$scene.json 2> $out/pre_$scene.err || exit 1
  timeout -k 10 900 bash bench/pmc.sh ${tag}_$scene --scene $scene $P > $out/pmc_$scene.log 2>&1 || exit 1
  cp gpurun_out/pmc_${tag}_$scene/summary.json $out/pmc_summary_${scene}_full_frame.json
  timeout -k 10 600 bash bench/pmc_stall.sh ${tag}_${scene}s --scene $scene $P > $out/pmc_ta_td_stall_$scene.txt 2> $out/stall_$scene.err || exit 1
  python bench/pmc_traffic_update.py $out/pmc_summary_${scene}_full_frame.json $scene "round 3 final source state, one full frame" > /dev/null &&
  python bench/pmc_stall_update.py $out/pmc_ta_td_stall_$scene.txt $scene $out/pre_$scene.json > /dev/null || exit 1
  echo "== $scene done"
done
cp profiles/pmc_traffic.json $out/pmc_traffic.json
timeout -k 10 200 python bench.py --scene cornell --steps 20 --warmup 3 > $out/bench_cornell.json 2> $out/bench_cornell.err
timeout -k 10 200 python bench.py --scene bunny --steps 5 --warmup 1 > $out/bench_bunny.json 2> $out/bench_bunny.err
timeout -k 10 600 python bench.py --scene glossy --steps 1 --warmup 0 > $out/bench_glossy.json 2> $out/bench_glossy.err
echo "== done"
