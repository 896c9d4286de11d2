#!/usr/bin/env python3
"""GPU box: wall-clock behaviour of whole plan() calls through the C++ host planner (path_planner_amd/host/plan_cli)
with a REAL clock and a fixed time budget: iterations, expansions and samples reached, and how far past the budget the
call returns.  usage: tools/plan_latency.py [cfg3] [budget_ms] [initial_samples] [repeat] [speculation]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_planner_amd import workloads
from test_gpu_host_planner import _write_map, _scenario, CLI

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
init = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 8
spec = int(sys.argv[5]) if len(sys.argv) > 5 else None
w = workloads.by_name(name)
with tempfile.TemporaryDirectory() as d:
    mp = os.path.join(d, "grid.map"); _write_map(w.grid, w.res, mp)
    sc = os.path.join(d, "s.txt")
    _scenario(w, sc, mp, 1000.0, 1e-3, 1, init, speculation=spec)
    with open(sc, "a") as f:
        f.write(f"time_remaining {budget / 1e3!r}\nreal_clock 1\nrepeat {repeat}\n")
    out = subprocess.run([CLI, sc], capture_output=True, text=True, timeout=600)
    print(out.stderr.strip().splitlines()[-1] if out.stderr.strip() else "")
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print("speculation", spec, {k: r[k] for k in ("samples", "expanded", "generated", "iterations", "first_goal_iteration", "edges_costed", "plan_f", "wall_ms_median", "wall_ms_max")})
