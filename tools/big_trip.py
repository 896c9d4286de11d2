#!/usr/bin/env python3
"""Developer tool (GPU box): what one ppgpu_expand_host round trip of 64 open vertices costs as the sample set doubles (8 192 ..
4 million samples), with config 3's ribbons and with ribbons that are almost covered (a late mission: the generator's ribbon
projections pile up on short pieces) — what the planner's deadline guard has to predict.  usage: tools/big_trip.py [nv] [k]
With PP_TRIP_TIMING=1 also the per-kernel HIP-event groups of the last trip."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from path_planner_amd import api, workloads
from path_planner_amd.types import VERTEX_DTYPE
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
w = workloads.config3()
w.obst = workloads.obstacles(32, 3, 204.8, time=float(w.start5[4]))
c = 102.4
cases = {"config 3's ribbons": w.ribbons4,
         "nearly covered (5 pieces of 2.5 m)": np.array([[c + 17.5, c + 12 + 8 * i, c + 20, c + 12 + 8 * i] for i in range(5)]),
         "one piece of 2.2 m": np.array([[c + 17.8, c + 12, c + 20, c + 12]])}
for name, rib in cases.items():
    ctx = api.Context(0)
    ctx.reserve_samples(8 << 20, 64)
    ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst)
    ctx.sampler_init(w.bounds6, w.seed, rib)
    rng = np.random.default_rng(5)
    print(f"== {name}: {nv} vertices, k = {k}")
    n_att = 8192
    have = 0
    while n_att <= (4 << 20):
        while have < n_att:
            step = min(n_att - have, 1 << 19); ns = ctx.sampler_add(step); have += step
        verts = np.zeros(nv, dtype=VERTEX_DTYPE)
        for i in range(nv):
            verts[i]["x"] = c + rng.uniform(-30, 30); verts[i]["y"] = c + rng.uniform(-10, 50); verts[i]["heading"] = rng.uniform(0, 2 * np.pi)
            verts[i]["speed"] = w.cfg.max_speed; verts[i]["time"] = float(w.start5[4]) + rng.uniform(0, 10); verts[i]["g"] = 1.0
            verts[i]["coverage_completed_time"] = -1.0; verts[i]["ribbon_offset"] = i * len(rib); verts[i]["ribbon_count"] = len(rib)
        ribs = np.tile(rib, (nv, 1))
        nearest = np.stack([np.full(nv, rib[0][0]), np.full(nv, rib[0][1]), np.zeros(nv)], axis=1)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            e, res, child = ctx.expand_host(verts, ribs, nearest, k, stride=13)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"   attempts {n_att:8d} samples {ns:8d}: trip {min(ts):8.3f} ms (first {ts[0]:8.3f})  edges {len(e):5d}  order fallbacks so far {ctx.order_fallbacks()}")
        n_att *= 2
    del ctx
