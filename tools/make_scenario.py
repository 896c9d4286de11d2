#!/usr/bin/env python3
"""Write a plan_cli scenario (+ map file) for config 3 to a directory.
usage: tools/make_scenario.py outdir [--cfg5] [--initial N] [extra line ...]
  --cfg5       32 moving obstacles uniform in the map instead of config 3's 16
  --initial N  initial samples (default 1024; config 5 says 8192)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_planner_amd import workloads
from test_gpu_host_planner import _write_map, _scenario
args = sys.argv[1:]
out = args.pop(0)
cfg5 = "--cfg5" in args
if cfg5:
    args.remove("--cfg5")
init = 1024
if "--initial" in args:
    i = args.index("--initial"); init = int(args[i + 1]); del args[i:i + 2]
os.makedirs(out, exist_ok=True)
w = workloads.config3()
if cfg5:
    w.obst = workloads.obstacles(32, 3, 204.8, time=float(w.start5[4]))
mp = os.path.join(out, "grid.map"); _write_map(w.grid, w.res, mp)
sc = os.path.join(out, "s.txt")
_scenario(w, sc, mp, float(w.start5[4]), 1e-3, 1, init)
with open(sc, "a") as f:
    for line in args:
        f.write(line + "\n")
print(sc)
