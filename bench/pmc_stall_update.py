"""Folds the TA/TD stall counters of bench/pmc_stall.sh (its printed table, saved as text) and the in-kernel lane-load count
of a bench.py line of the same frame into profiles/pmc_traffic.json: the fractions bench.py reports beside the roofline.

    python bench/pmc_stall_update.py profiles/r03/final/pmc_ta_td_stall_sponza.txt sponza [bench.json of the same frame]

ta_busy_frac            = TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs): share of the CUs' cycles the texture-address unit is busy
ta_stalled_by_tc_frac   = TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_TA_BUSY_sum: share of those busy cycles spent stalled by the cache behind it
td_busy_frac, td_stalled_frac: the same for the texture-data unit (TD_TD_BUSY_sum, TD_TC_STALL_sum)
tcp_accesses_per_issued_lane_load = TCP_TOTAL_CACHE_ACCESSES (pmc_traffic.json entry, same kernel hash) / lane-loads counted in the kernel
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_CU, N_XCD = 256, 8
src, scene = sys.argv[1], sys.argv[2]
bench_json = sys.argv[3] if len(sys.argv) > 3 else None
c = {}
for line in open(src):
    parts = line.split()
    if len(parts) == 2:
        try:
            c[parts[0]] = float(parts[1])        # (GRBM_GUI_ACTIVE appears once per pass: the last one stands)
        except ValueError:
            pass
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
allp = json.load(open(path))
e = allp[scene]
cu_cycles = c["GRBM_GUI_ACTIVE"] / N_XCD * N_CU
e["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / cu_cycles
e["ta_stalled_by_tc_frac"] = c["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / c["TA_TA_BUSY_sum"]
e["td_busy_frac"] = c["TD_TD_BUSY_sum"] / cu_cycles
e["td_stalled_frac"] = c["TD_TC_STALL_sum"] / c["TD_TD_BUSY_sum"]
e["tcp_tagconflict_stall_frac_of_cu_cycles"] = c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0.0) / cu_cycles
e["stall_source"] = os.path.relpath(os.path.abspath(src), ROOT)
if bench_json:
    b = json.load(open(bench_json))
    r = b["roofline"]
    counted = r.get("useful_lane_loads_per_launch") or r.get("lane_loads_per_launch")
    if counted and r.get("kernel_source_hash") == e.get("kernel_source_hash"):
        e["issued_lane_loads_per_launch"] = counted
        e["tcp_accesses_per_issued_lane_load"] = e["tcp_lane_accesses_per_launch"] / counted
        e["lane_load_source"] = os.path.relpath(os.path.abspath(bench_json), ROOT)
json.dump(allp, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(e, indent=1, sort_keys=True))
