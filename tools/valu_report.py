#!/usr/bin/env python3
"""gpurun_out/valu.json from the two PMC passes of tools/valu.sh: how busy the vector ALU is in the costing kernels, from counters
(not from the survey's flop model).  Per kernel, per launch (mean over the launches measured):
  wave_insts_valu      SQ_INSTS_VALU: wave64 VALU instructions issued
  valu_busy_cycles     4 x SQ_ACTIVE_INST_VALU (the SQ counts quad-cycles, MI355X_MICROARCH.md cycle constants): cycles, summed over
                       the kernel's waves, in which a VALU instruction of the wave was issuing
  shader_cycles        GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs): cycles the dispatch was on the machine (second pass)
  valu_issue_frac      valu_busy_cycles / (1 024 SIMDs x shader_cycles): the share of the machine's VALU issue capacity used
The launch's figure weights the kernels by their shader cycles (pp_k_heuristic_listed runs beside pp_k_heuristic_lanes on a second
stream: its cycles overlap and are left out of the denominator, its instructions stay in the numerator).  executed_lane_ops =
wave_insts_valu x 64: what bench.py divides by the survey's algorithmic flops (an upper bound of the lanes doing arithmetic)."""
import collections, csv, glob, hashlib, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COSTING = ("pp_k_solve_edges", "pp_k_plan_skips", "pp_k_pose_sweep", "pp_k_approach_events", "pp_k_cover_sweep", "pp_k_cover_finish", "pp_k_deferred_list",
           "pp_k_heuristic_lanes", "pp_k_heuristic_listed", "pp_k_heuristic_big")
OVERLAPPED = ("pp_k_heuristic_listed",)
N_SIMD = 256 * 4


def per_kernel(dirname, counters):
    f = max(glob.glob(os.path.join(ROOT, "gpurun_out", dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)
    tot = collections.defaultdict(collections.Counter)
    calls = collections.defaultdict(set)
    ns = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k not in COSTING or r["Counter_Name"] not in counters:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[k]:
            calls[k].add(r["Dispatch_Id"])
            ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: {**{c: tot[k][c] / len(calls[k]) for c in counters}, "ns": ns[k] / len(calls[k]), "launches_measured": len(calls[k])} for k in tot}


sq = per_kernel("valu_sq", ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"))
clk = per_kernel("valu_clk", ("GRBM_GUI_ACTIVE",))
out = {"unit": "per kernel launch (mean over the launches measured)", "kernels": {},
       "note": "SQ counters count quad-cycles (x4); GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8); 1 024 SIMDs"}
busy = cyc = insts = 0.0
for k in COSTING:
    if k not in sq or k not in clk:
        continue
    s, c = sq[k], clk[k]
    shader = c["GRBM_GUI_ACTIVE"] / 8.0
    vb = 4.0 * s["SQ_ACTIVE_INST_VALU"]
    out["kernels"][k] = {"waves": s["SQ_WAVES"], "wave_insts_valu": s["SQ_INSTS_VALU"], "valu_insts_per_wave": s["SQ_INSTS_VALU"] / max(s["SQ_WAVES"], 1.0),
                         "valu_busy_cycles": vb, "wave_cycles": 4.0 * s["SQ_WAVE_CYCLES"], "shader_cycles": shader,
                         "clock_ghz": shader / max(c["ns"], 1.0), "us": c["ns"] / 1e3,
                         "valu_issue_frac": vb / (N_SIMD * shader) if shader > 0 else None, "launches_measured": s["launches_measured"]}
    busy += vb
    insts += s["SQ_INSTS_VALU"]
    if k not in OVERLAPPED:
        cyc += shader
out["valu_busy_cycles_per_launch"] = busy
out["shader_cycles_per_launch"] = cyc
out["valu_issue_frac"] = busy / (N_SIMD * cyc) if cyc > 0 else None
out["wave_insts_valu_per_launch"] = insts
out["executed_lane_ops_per_launch"] = insts * 64.0
h = hashlib.sha256()
import sys
sys.path.insert(0, ROOT)
from bench import KERNEL_SOURCES
for f in KERNEL_SOURCES:
    h.update(open(os.path.join(ROOT, "path_planner_amd", "csrc", f), "rb").read())
out["kernel_sources_sha256"] = h.hexdigest()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "valu.json"), "w"), indent=1)
for k, d in out["kernels"].items():
    print(f"{k:26s} {d['us']:8.1f} us  {d['clock_ghz']:5.2f} GHz  VALU/wave {d['valu_insts_per_wave']:9.0f}  issue share {d['valu_issue_frac']:.3f}")
print(f"launch: VALU issue share {out['valu_issue_frac']:.3f}, {insts:.3e} wave-instructions")
