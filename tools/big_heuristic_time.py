#!/usr/bin/env python3
"""Developer tool (GPU box): ppgpu_heuristic_host on lists of 9 .. 12 ribbons under TspPointRobotNoSplitKRibbons (K = 2): what
pp_k_heuristic_big takes per list (PPGPU_LIB_OVERRIDE selects a variant library)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from path_planner_amd import api
from path_planner_amd.types import make_config, H_TSP_POINT_K
rng = np.random.default_rng(3)
cfg = make_config(heuristic=H_TSP_POINT_K, tsp_k=2, ribbon_width=2.0, max_speed=2.0, heuristic_turning_radius=6.0)
ctx = api.Context(0)
ctx.set_config(cfg)
for n in (9, 10, 11, 12):
    for layout in ("random", "survey lines"):
        poses, lists = [], []
        for _ in range(8):
            if layout == "random":
                lists.append(rng.uniform(0, 120, (n, 4)))
            else:
                y = 10.0 + 6.0 * np.arange(n)
                lists.append(np.stack([np.full(n, 20.0), y, np.full(n, 100.0), y], axis=1) + rng.uniform(-0.5, 0.5, (n, 4)))
            poses.append([rng.uniform(0, 120), rng.uniform(0, 120), rng.uniform(0, 2 * np.pi)])
        g, fl = ctx.heuristic_host(poses, lists)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); g2, _ = ctx.heuristic_host(poses, lists); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{os.environ.get('PPGPU_LIB_OVERRIDE', 'default'):28s} n {n:2d} {layout:12s}: 8 lists in {min(ts):8.3f} ms   h[0] {g[0]!r}")
