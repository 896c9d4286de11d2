#!/usr/bin/env python3
"""profiles/traffic.json from the two PMC passes of tools/traffic.sh.  Per kernel and per launch: FETCH_SIZE and
WRITE_SIZE (rocprofv3 reports KiB-scaled units of 1 KB = 1024 B per the counter definitions; TCC_EA request counts x 64 B),
with the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 128-B read requests as 64 B, so reads are
doubled.  The guide states that for 16-byte-per-lane coalesced streaming reads and calls other shapes uncalibrated;
tools/fetch_calibrate.sh measured this library's own shapes on MI355X (profiles/r03_fetch_calibration.json): coalesced
8-byte-per-lane reads and one-lane-per-384-byte-record gathers that touch all three lines of a record ALSO read 0.50 x the
bytes fetched, so the factor 2 is applied to every kernel; WRITE_SIZE is exact for coalesced 8-byte stores and counts 32 B per
8-byte store scattered 384 bytes apart (sector granularity: real traffic, not useful bytes).
`hbm_bytes_per_launch` is the dominant kernel's (pp_k_cover_sweep) corrected read + write bytes;
`hbm_bytes_per_costing_launch` sums the kernels of one ppgpu_cost_edges_* launch (solve .. heuristics)."""
import collections, csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"unit": "bytes per kernel launch",
       "correction": "FETCH_SIZE x2 for every kernel (gfx950 counts a 128-B read request as 64 B; calibrated on this library's access shapes: "
                     "16 B/lane and 8 B/lane coalesced reads and 384-B-stride record gathers all read 0.50 x, profiles/r03_fetch_calibration.json); "
                     "WRITE_SIZE x1 (exact for coalesced stores; 32 B per scattered 8-byte store); counter unit 1 KB = 1024 B",
       "kernels": {}}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(os.path.join(ROOT, "gpurun_out", f"traffic_{counter}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    tot, calls = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); calls[k] += 1
    for k in tot:
        if not k.startswith("pp_k_"):
            continue
        d = out["kernels"].setdefault(k, {})
        scale = 2.0 if counter == "FETCH_SIZE" else 1.0
        d[counter.lower() + "_bytes"] = tot[k] / calls[k] * 1024.0 * scale
        d["launches_measured"] = calls[k]
dom = out["kernels"].get("pp_k_cover_sweep", {})
out["dominant_kernel"] = "pp_k_cover_sweep"
out["hbm_bytes_per_launch"] = dom.get("fetch_size_bytes", 0.0) + dom.get("write_size_bytes", 0.0)
COSTING = ("pp_k_solve_edges", "pp_k_plan_skips", "pp_k_pose_sweep", "pp_k_approach_events", "pp_k_cover_sweep", "pp_k_cover_finish", "pp_k_deferred_list",
           "pp_k_heuristic_lanes", "pp_k_heuristic_listed", "pp_k_heuristic_big")
out["costing_kernels"] = list(COSTING)
out["hbm_bytes_per_costing_launch"] = sum(out["kernels"].get(k, {}).get("fetch_size_bytes", 0.0) + out["kernels"].get(k, {}).get("write_size_bytes", 0.0) for k in COSTING)
import hashlib
_h = hashlib.sha256()
import sys
sys.path.insert(0, ROOT)
from bench import KERNEL_SOURCES
for _f in KERNEL_SOURCES:
    _h.update(open(os.path.join(ROOT, "path_planner_amd", "csrc", _f), "rb").read())
out["kernel_sources_sha256"] = _h.hexdigest()       # bench.py reports these bytes only while the sources are the ones measured
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
for k, d in sorted(out["kernels"].items()):
    print(f"{k:28s} read {d.get('fetch_size_bytes', 0) / 1e6:10.1f} MB   write {d.get('write_size_bytes', 0) / 1e6:10.1f} MB   per launch")
