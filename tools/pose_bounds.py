#!/usr/bin/env python3
"""Developer tool (GPU box): what the collision sweep (pp_k_plan_skips + pp_k_pose_sweep) of the bench workload costs with and
without its dynamic obstacles and with an empty grid — upper bounds on what any treatment of obstacle-boundary chunks, or of chunks
near blocked cells, can gain."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_planner_amd import api, workloads
w = workloads.config3()
def run(label, grid, obst):
    ctx = api.Context(0)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
    ctx.set_config(w.cfg); ctx.set_grid(grid, w.res); ctx.set_obstacles(obst); ctx.set_vertices(w.root(), w.ribbons4)
    ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
    d = torch.zeros(4 * n * 128, dtype=torch.uint8, device="cuda")
    ctx.enable_timing(True)
    ts = []
    for i in range(6):
        ctx.cost_edges_dense(0, 1, 0, n, 0xF, d.data_ptr()); ts.append(ctx.last_timing())
    print(f"{label:34s} edges {4 * n:7d}  solve/pose/cover/heur ms:", np.round(np.median(np.array(ts[1:]), axis=0), 3), flush=True)
run("config 3", w.grid, w.obst)
run("no dynamic obstacles", w.grid, None)
run("empty grid (same samples kept? no)", np.zeros_like(w.grid), w.obst)
run("empty grid, no obstacles", np.zeros_like(w.grid), None)
