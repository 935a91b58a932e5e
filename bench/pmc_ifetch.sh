#!/bin/bash
# Instruction-fetch side of render_kernel (one rocprofv3 pass).  Usage: bash bench/pmc_ifetch.sh <tag> [bench.py args...]
tag=$1; shift
args=${@:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-count}
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 180 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out -o if1 -- python bench.py $args > $out/if1.log 2>&1 || { echo "pass if1 failed"; tail -5 $out/if1.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES --output-format csv -d $out -o if2 -- python bench.py $args > $out/if2.log 2>&1 || { echo "pass if2 failed"; tail -5 $out/if2.log; exit 1; }
python - $out <<'P'
import csv, sys, collections
for name in ("if1", "if2"):
    tot = collections.defaultdict(float)
    with open(f"{sys.argv[1]}/{name}_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if "render_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()): print(f"{k:32s} {v:.4g}")
P
