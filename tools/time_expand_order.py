#!/usr/bin/env python3
"""GPU box (run under rocprofv3 --kernel-trace --stats): the expand-order pipeline on 16 open vertices at several sample counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE, VERTEX_DTYPE, F_INFEASIBLE
n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
w = workloads.config3()
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(n_samples)
e = np.arange(64, dtype=np.uint64) * 4
res, child = ctx.cost_edges_host(e, stride=12)
ok = np.nonzero((res["flags"] & F_INFEASIBLE) == 0)[0][:15]
v = np.zeros(len(ok) + 1, dtype=VERTEX_DTYPE); pool = [np.asarray(w.ribbons4).reshape(-1, 4)]; v[0] = w.root()[0]; off = len(pool[0])
for k, i in enumerate(ok):
    r, nr = res[i], int((res[i]["info"] >> 8) & 0xFF)
    v[k + 1] = (r["end_x"], r["end_y"], r["end_heading"], r["end_speed"], r["end_time"], r["g"], r["coverage_completed_time"], off, nr)
    pool.append(child[i, :nr]); off += nr
ctx.set_vertices(v, np.concatenate(pool))
for _ in range(10):
    idx, fb = ctx.expand_order(len(v), 9)
print("samples", n, "vertices", len(v), "fallbacks", fb)
