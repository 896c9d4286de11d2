#!/usr/bin/env python3
"""CPU: search for fixed-seed planner scenarios whose FIRST GOAL comes late (oracle first_goal_iteration >= 2): heavily blocked
grids and few initial samples, so that the first iterations exhaust the open list without reaching the horizon.  Prints the
parameters of the hits; tests/test_gpu_host_planner.py pins some of them.  usage: tools/find_late_goal.py [cfg2|cfg3] [tries]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as orc
from path_planner_amd import workloads

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 40
base = workloads.by_name(which)
n = base.grid.shape[0]
c = float(base.start5[0])
hits = []
for t in range(tries):
    frac = [0.2, 0.25, 0.3, 0.35][t % 4]
    gseed = 100 + t
    init = [8, 16, 32][(t // 4) % 3]
    blob = [8, 12][(t // 12) % 2] if which == "cfg2" else [16, 24][(t // 12) % 2]
    grid = workloads.blob_grid(n, base.res, frac, gseed, (c, c), keep_free_radius=2.0, blob=blob)
    orc.O.ppo_set_ribbon_width(base.cfg.ribbon_width)
    world = orc.World(base.cfg, grid, base.res, base.obst)
    rc, st, plan, itf, _ = world.plan(base.ribbons4, base.start5, 400 * 1e-3, 1000.0, 1e-3, initial_samples=init)
    print(which, "frac", frac, "grid_seed", gseed, "init", init, "blob", blob, "->", "first_goal", st.first_goal_iteration, "iters", st.iterations,
          "expanded", st.expanded, "samples", st.samples, "plan_len", st.plan_len, flush=True)
    if st.first_goal_iteration >= 2:
        hits.append((frac, gseed, init, blob, st.first_goal_iteration))
print("hits:", hits)
