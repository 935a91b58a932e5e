"""Folds a PMC summary (bench/pmc_summary.py, made on the GPU box by bench/pmc.sh) into profiles/pmc_traffic.json, the
file bench.py reads for `roofline.traffic` and the other PMC-derived fields.  The entry keeps the kernel-source
hash the passes were measured at; bench.py drops the entry when the kernel has changed since.

    python bench/pmc_traffic_update.py profiles/r02/pmc_summary_sponza.json sponza "one 256-spp launch ..."
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, scene = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
a = json.load(open(src))
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
allp = json.load(open(path)) if os.path.exists(path) else {}
launches = max(int(a.get("launches", 1)), 1)
allp[scene] = {
    "kernel_source_hash": a["kernel_source_hash"],
    "source": os.path.relpath(os.path.abspath(src), ROOT),
    "launches_summed": launches,
    "hbm_bytes_per_launch": a["hbm_bytes_per_launch"],
    "read_bytes_raw_fetch_size": a["hbm_read_bytes_raw"] / launches,
    "read_bytes_x2_gfx950_correction": a["hbm_read_bytes_x2"] / launches,
    "write_bytes": a["hbm_write_bytes"] / launches,
    "tcp_lane_accesses_per_launch": a["TCP_TOTAL_CACHE_ACCESSES_sum"] / launches,
    "l2_hit_rate": a["l2_hit_rate"], "l1_miss_to_l2_frac": a["l1_miss_to_l2_frac"],
    "valu_busy_frac_of_cu_time": a["valu_busy_frac_of_cu_time"], "lane_utilisation": a["lane_utilisation"],
    "wait_any_frac": a["wait_any_frac"],
    "note": note or "rocprofv3 --pmc, one run per pass (bench/pmc.sh); FETCH_SIZE x2 per MI355X_MICROARCH.md (calibrated for this "
                    "gather pattern in profiles/r01_tuning.md), fabric-side requests, Infinity-Cache hits included",
}
json.dump(allp, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(allp[scene], indent=1))
