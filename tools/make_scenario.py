#!/usr/bin/env python3
"""Write a plan_cli scenario (+ map file) for config 3 to a directory.  usage: tools/make_scenario.py outdir [extra line ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_planner_amd import workloads
from test_gpu_host_planner import _write_map, _scenario
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
w = workloads.config3()
mp = os.path.join(out, "grid.map"); _write_map(w.grid, w.res, mp)
sc = os.path.join(out, "s.txt")
_scenario(w, sc, mp, float(w.start5[4]), 1e-3, 1, 1024)
with open(sc, "a") as f:
    for line in sys.argv[2:]:
        f.write(line + "\n")
print(sc)
