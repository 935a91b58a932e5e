#!/bin/bash
# In-flight and latency counters of render_kernel on the tree that streams from HBM (C4), at several blocks per CU:
# what caps the 6.2 TB/s of fabric traffic -- misses in flight or their latency.
# Usage: bash bench/pmc_c4.sh <tag> [blocks per CU ...]      -> gpurun_out/pmc_c4_<tag>/table.txt
tag=$1; shift
bpcs=${@:-5 6 7}
args="--scene tenm --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-count --no-iter"
out=gpurun_out/pmc_c4_$tag
mkdir -p $out
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python bench.py $args > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
for b in $bpcs; do
  export HPT_BLOCKS_PER_CU=$b
  run ea_b$b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE &&
  run l2_b$b TCC_TAG_STALL_sum TCC_REQ_sum TCC_MISS_sum TCC_LATENCY_FIFO_FULL_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
done
python - $out $bpcs <<'P' | tee $out/table.txt
import csv, sys, collections, os
out, bpcs = sys.argv[1], sys.argv[2:]
N_CU, N_XCD, CLK = 256, 8, 2.4e9
print("%-4s %9s %12s %12s %11s %11s %12s %12s %12s %11s" % ("b/CU", "kernel ms", "EA rd reqs", "EA in flight", "EA lat clk", "EA lat ns", "TB/s @128B", "L1 miss lat", "L1 misses", "VMEM lat"))
for b in bpcs:
    c = collections.defaultdict(float)
    for name in ("ea_b" + b, "l2_b" + b):
        path = os.path.join(out, name + "_counter_collection.csv")
        for r in csv.DictReader(open(path)):
            if "render_kernel" in r["Kernel_Name"]:
                c[r["Counter_Name"]] += float(r["Counter_Value"])
    cycles = c["GRBM_GUI_ACTIVE"] / N_XCD / 2        # two passes each add theirs
    ms = cycles / CLK * 1e3
    ea = c["TCC_EA0_RDREQ_sum"]
    inflight = c["TCC_EA0_RDREQ_LEVEL_sum"] / cycles          # mean read requests in flight at the fabric interface, whole chip
    lat = c["TCC_EA0_RDREQ_LEVEL_sum"] / max(ea, 1)           # Little: level-cycles per request
    tbs = ea * 128 / (ms * 1e-3) / 1e12                       # every request of a scattered 64-B read fills a 128-B line (r01 calibration)
    l1lat = c["TCP_TCC_READ_REQ_LATENCY_sum"] / max(c["TCP_TCC_READ_REQ_sum"], 1)
    vlat = c["SQ_INST_LEVEL_VMEM"] / max(c["SQ_INSTS_VMEM_RD"], 1)
    print("%-4s %9.1f %12.4g %12.0f %11.0f %11.0f %12.2f %12.0f %12.4g %11.0f" % (b, ms, ea, inflight, lat, lat / CLK * 1e9, tbs, l1lat, c["TCP_TCC_READ_REQ_sum"], vlat))
    print("     per CU: L1 misses in flight %.1f (misses x latency / CU cycles); DRAM credit stall %.3g, TCC tag stall %.3g, TCC busy %.3g of %.3g channel-cycles; pending-stall %.3g"
          % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / (cycles * N_CU), c["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"], c["TCC_TAG_STALL_sum"], c["TCC_BUSY_sum"], cycles * 16 * N_XCD, c["TCP_PENDING_STALL_CYCLES_sum"]))
P
