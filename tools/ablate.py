#!/usr/bin/env python3
"""Developer tool (GPU box): build variants of libppgpu.so with parts of the edge kernel compiled out
and time the costing launch of the bench workload with each, to see where the time goes.
Usage: python tools/ablate.py [variant=flags ...]"""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "abl")
VARIANTS = {
    "full": [],
    "norun": ["-DPP_NO_CORRIDOR_RUN"],
    "noquiet": ["-DPP_NO_QUIET_RUN"],
    "h3": ["-DPP_H_MIN_WAVES=3"],
    "h4": ["-DPP_H_MIN_WAVES=4"],
    "h5": ["-DPP_H_MIN_WAVES=5"],
    "h6": ["-DPP_H_MIN_WAVES=6"],
    "h7": ["-DPP_H_MIN_WAVES=7"],
    "h8": ["-DPP_H_MIN_WAVES=8"],
    "occ1": ["-DPP_MIN_WAVES=1"],
    "occ4": ["-DPP_MIN_WAVES=4"],
    "occ5": ["-DPP_MIN_WAVES=5"],
    "occ6": ["-DPP_MIN_WAVES=6"],
    "occ8": ["-DPP_MIN_WAVES=8"],
    "pose4": ["-DPP_POSE_MIN_WAVES=4"],
    "pose6": ["-DPP_POSE_MIN_WAVES=6"],
    "pose7": ["-DPP_POSE_MIN_WAVES=7"],
    "pose8": ["-DPP_POSE_MIN_WAVES=8"],
    "no_track_store": ["-DPP_ABL_NO_TRACK_STORE"],
    "wpb1": ["-DPP_WPB=1"],
    "wpb2": ["-DPP_WPB=2"],
    "wpb8": ["-DPP_WPB=8"],
    "no_heur": ["-DPP_ABL_NO_HEUR"],
    "no_obst": ["-DPP_ABL_NO_OBST"],
    "no_events": ["-DPP_ABL_NO_EVENTS"],
    "no_grid": ["-DPP_ABL_NO_GRID"],
    "no_heur_obst": ["-DPP_ABL_NO_HEUR", "-DPP_ABL_NO_OBST"],
    "no_heur_obst_events": ["-DPP_ABL_NO_HEUR", "-DPP_ABL_NO_OBST", "-DPP_ABL_NO_EVENTS"],
    "no_heur_obst_events_grid": ["-DPP_ABL_NO_HEUR", "-DPP_ABL_NO_OBST", "-DPP_ABL_NO_EVENTS", "-DPP_ABL_NO_GRID"],
}

CHILD = r'''
import os, sys, json, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE
w = workloads.config3(n_samples=%(n)d)
ctx = api.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4*n*128, dtype=torch.uint8, device="cuda")
ts = []
ks = []
ctx.enable_timing(True)
for i in range(8):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(st); ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); b.record(st); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b)); ks.append(ctx.last_timing())
print(json.dumps({"ms": min(ts[1:]), "edges": 4*n, "kernels": [float(x) for x in np.median(np.array(ks[1:]), axis=0)]}))
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    names = sys.argv[1:] or list(VARIANTS)
    n = int(os.environ.get("ABL_SAMPLES", "65536"))
    for name in names:
        flags = VARIANTS[name] if name in VARIANTS else name.split("=", 1)[1].split(",")
        name = name.split("=", 1)[0]
        lib = os.path.join(OUT, f"libppgpu_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
                               "-std=c++17"] + flags + [os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
        env = dict(os.environ, PPGPU_LIB_OVERRIDE=lib)
        out = subprocess.check_output([sys.executable, "-c", CHILD % {"root": ROOT, "n": n}], env=env).decode().strip().splitlines()[-1]
        r = json.loads(out)
        print(f"{name:28s} {r['ms']:9.3f} ms   {r['edges'] / r['ms'] / 1e3:8.2f} Medges/s   solve/pose/cover/heuristic " +
              " ".join(f"{x:6.3f}" for x in r["kernels"]), flush=True)


if __name__ == "__main__":
    main()
