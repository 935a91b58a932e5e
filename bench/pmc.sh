#!/bin/bash
# PMC passes of bench.py's render kernel, one rocprofv3 run per pass (slots: 8 SQ, 4 TCC, 4 TCP, 2 GRBM;
# FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  Usage (on the GPU box, from the repo root):
#   bash bench/pmc.sh <tag> [bench.py args...]        -> gpurun_out/pmc_<tag>/<pass>_counter_collection.csv
tag=$1; shift
args=${@:---steps 1 --warmup 0 --no-cpu-baseline --no-count}
out=gpurun_out/pmc_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python bench.py $args > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; return 1; }
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA &&
run mem TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE GRBM_COUNT &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
python bench/pmc_summary.py $out
