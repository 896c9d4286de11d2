#!/usr/bin/env python3
"""GPU box: SURVEY.md config 5 on one GPU — a 10 Hz anytime replan loop through the C++ host planner: moving start, 32
moving obstacles, 100 ms budget per cycle, previous plan handed back each cycle.  Prints p50/p99 plan() wall time,
iterations, expansions and samples reached per cycle.  usage: tools/replan_loop.py [cycles] [budget_ms] [initial_samples] [speculation, 0 = default] [streams]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from path_planner_amd import workloads
from test_gpu_host_planner import _write_map, _scenario, CLI

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
init = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
spec = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else None
streams = int(sys.argv[5]) if len(sys.argv) > 5 else 2       # device contexts (HIP streams) on the one GPU
w = workloads.config3()
w.obst = workloads.obstacles(32, int(os.environ.get("REPLAN_OBST_SEED", "3")), 204.8, time=float(w.start5[4]))      # uniform in the map: no free disc around the start
with tempfile.TemporaryDirectory() as d:
    mp = os.path.join(d, "grid.map"); _write_map(w.grid, w.res, mp)
    sc = os.path.join(d, "s.txt")
    _scenario(w, sc, mp, float(w.start5[4]), 1e-3, 1, init, speculation=spec, devices=[0] * streams)
    with open(sc, "a") as f:
        f.write(f"time_remaining {budget / 1e3!r}\nreplan {cycles} 0.1\n")
        if os.environ.get("REPLAN_CYCLE_LOG"):
            f.write(f"cycle_log {os.environ['REPLAN_CYCLE_LOG']}\n")
    out = subprocess.run([CLI, sc], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    sys.stderr.write("".join(l + "\n" for l in out.stderr.splitlines() if l.startswith("[replan]") or l.startswith("[profile]")))
    r = json.loads(out.stdout.strip().splitlines()[-1])
    r.update({"initial_samples": init, "speculation": spec if spec is not None else 64, "streams": streams, "obstacles": 32, "workload": "cfg3 grid, 5 ribbons"})
    print(json.dumps(r))
