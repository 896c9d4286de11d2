#!/usr/bin/env python3
"""GPU box, developer aid: one round of tools/fuzz_plan.py in detail — both sides' statistics and where the two edge dumps part.
usage: tools/dbg_plan_round.py <seed> <round> [replan]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_plan as fp
import oracle as orc
from test_gpu_host_planner import _write_map, _scenario, _run_cli

seed, rid = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for r in range(rid):
    fp.make_round(rng, r)
w, t0, dt, calls, init, spec, tag = fp.make_round(rng, rid)
print(tag, "speculation", spec)
cfg = w.cfg
orc.O.ppo_set_ribbon_width(cfg.ribbon_width)
world = orc.World(cfg, w.grid, w.res, w.obst)
with tempfile.TemporaryDirectory() as d:
    mp = os.path.join(d, "grid.map"); _write_map(w.grid, w.res, mp)
    sc = os.path.join(d, "s.txt")
    for sp in (spec, 1):
        _scenario(w, sc, mp, t0, dt, calls, init, speculation=sp)
        host = _run_cli(sc)
        print("host (speculation %d):" % sp, {k: host.get(k) for k in ("samples", "iterations", "expanded", "generated", "first_goal_iteration", "plan_f", "depth", "host_heuristics", "order_fallbacks", "exception")})
    rc, st, plan, _, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init)
    print("oracle:", rc, {k: getattr(st, k) for k in ("samples", "iterations", "expanded", "generated", "first_goal_iteration", "plan_f")})
    H, O = fp.edge_dumps(w, sc, mp, t0, dt, calls, init, world)
    print("edges consumed: host", len(H), "oracle", len(O))
    rel = lambda a, b: np.abs(a - b) / np.maximum(1.0, np.abs(b))
    for i in range(min(len(H), len(O))):
        if np.max(rel(H[i], O[i])) > 1e-9:
            print("first difference at edge", i)
            for j in range(max(0, i - 2), min(len(H), len(O), i + 6)):
                print("  H", j, np.array2string(H[j], precision=12, max_line_width=400))
                print("  O", j, np.array2string(O[j], precision=12, max_line_width=400))
            break
    else:
        print("the common prefix of the dumps is identical; lengths", len(H), len(O))
        n = min(len(H), len(O))
        for j in range(n, min(max(len(H), len(O)), n + 4)):
            print("  extra", "H" if len(H) > len(O) else "O", j, np.array2string((H if len(H) > len(O) else O)[j], precision=10, max_line_width=400))
    _scenario(w, sc, mp, t0, dt, calls, init, speculation=spec)
    host = _run_cli(sc)
    hp = np.array(host["plan"], dtype=np.float64).reshape(-1, 11)
    print("host plan depth", host.get("plan_depth"), "oracle", st.plan_depth, "plan_h", host.get("plan_h"), st.plan_h)
    for i in range(max(len(hp), len(plan))):
        if i < len(hp): print("  H", i, np.array2string(hp[i], precision=15, max_line_width=400))
        if i < len(plan): print("  O", i, np.array2string(plan[i], precision=15, max_line_width=400))
