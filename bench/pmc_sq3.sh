#!/bin/bash
# Issue-side detail of render_kernel (one rocprofv3 pass).  Usage: bash bench/pmc_sq3.sh <tag> [bench.py args...]
tag=$1; shift
args=${@:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-count}
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 180 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $out -o sq3 -- python bench.py $args > $out/sq3.log 2>&1 || { echo "pass sq3 failed"; tail -5 $out/sq3.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL --output-format csv -d $out -o sq4 -- python bench.py $args > $out/sq4.log 2>&1 || { echo "pass sq4 failed"; tail -5 $out/sq4.log; exit 1; }
python - $out <<'P'
import csv, sys, collections
for name in ("sq3", "sq4"):
    tot = collections.defaultdict(float)
    with open(f"{sys.argv[1]}/{name}_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if "render_kernel" in r["Kernel_Name"] or "render2_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()): print(f"{k:32s} {v:.4g}")
P
