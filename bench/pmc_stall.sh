#!/bin/bash
# Address-translation and TA/TCP stall counters of render_kernel (three rocprofv3 passes).
# Usage: bash bench/pmc_stall.sh <tag> [bench.py args...]      -> gpurun_out/pmc_<tag>/stall.txt (bench/pmc_stall_update.py folds it
# into profiles/pmc_traffic.json)
tag=$1; shift
args=${@:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-count}
out=gpurun_out/pmc_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python bench.py $args > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
run st1 TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE &&
run st2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE &&
run st3 TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS GRBM_GUI_ACTIVE
python - $out <<'P' | tee $out/stall.txt
import csv, sys, collections, os
for name in ("st1", "st2", "st3"):
    path = f"{sys.argv[1]}/{name}_counter_collection.csv"
    if not os.path.exists(path): continue
    tot = collections.defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            if "render_kernel" in r["Kernel_Name"] or "render2_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()): print(f"{k:44s} {v:.4g}")
P
