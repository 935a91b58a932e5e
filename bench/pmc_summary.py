"""Sums the PMC passes of bench/pmc.sh over the render_kernel dispatches and derives the ratios used in
DESIGN.md (lane utilisation, wait fraction, L2 hit rate, HBM-side traffic with the gfx950 FETCH_SIZE
correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (bench.py at the repo root)

d = sys.argv[1]
agg = collections.defaultdict(float)
launches = 0
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"] or "render2_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
    launches = max(launches, len(seen))
a = dict(agg)
# "launch" in the roofline = one hpt_render call = one frame, which the backend may cut into several dispatches of
# render_kernel (sample buffer); the PMC passes run bench.py for PMC_FRAMES frames (default 1)
a["dispatches"] = launches
launches = int(os.environ.get("PMC_FRAMES", "1"))
a["launches"] = launches
a["kernel_source_hash"] = kernel_source_hash()      # bench.py reports these figures only while the kernel is this one
g = a.get
if g("SQ_ACTIVE_INST_VALU"):
    a["lane_utilisation"] = a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64)
    a["valu_active_frac_of_wave_cycles"] = a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]
    a["wait_any_frac"] = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]
    a["valu_busy_frac_of_cu_time"] = a["SQ_ACTIVE_INST_VALU"] / (a["SQ_BUSY_CYCLES"] * 4 * 4) if g("SQ_BUSY_CYCLES") else None
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    a["l2_hit_rate"] = a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a["TCC_MISS_sum"], 1)
if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    a["l1_miss_to_l2_frac"] = a["TCP_TCC_READ_REQ_sum"] / a["TCP_TOTAL_CACHE_ACCESSES_sum"]
    a["mean_l2_read_latency_cycles"] = a["TCP_TCC_READ_REQ_LATENCY_sum"] / max(a["TCP_TCC_READ_REQ_sum"], 1)
if g("FETCH_SIZE") is not None:
    # FETCH_SIZE is in KiB; on gfx950 it reports half the bytes of wide (128-B request) reads -> x2 per the guide.
    # This kernel's reads are 16 B/lane scattered (64-B nodes, 48-B triangles), a pattern the guide calls
    # uncalibrated: both the raw and the corrected figure are kept.
    a["hbm_read_bytes_raw"] = a["FETCH_SIZE"] * 1024
    a["hbm_read_bytes_x2"] = a["FETCH_SIZE"] * 2048
if g("WRITE_SIZE") is not None:
    a["hbm_write_bytes"] = a["WRITE_SIZE"] * 1024
if launches and g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    a["hbm_bytes_per_launch"] = (a["hbm_read_bytes_x2"] + a["hbm_write_bytes"]) / launches
json.dump(a, open(os.path.join(d, "summary.json"), "w"), indent=1, sort_keys=True)
for k in sorted(a):
    print("%-40s %s" % (k, a[k]))
