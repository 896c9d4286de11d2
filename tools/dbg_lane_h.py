#!/usr/bin/env python3
"""GPU box, developer aid: lane heuristic vs wave heuristic on one of test_lane_heuristic_is_the_wave_heuristic's cases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api
from path_planner_amd.types import H_TSP_POINT_ALL, H_TSP_POINT_K, make_config
from path_planner_amd.workloads import root_vertex
from test_gpu_parity import _dense
heur, k, nrib = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(100 * nrib + k)
cfg = make_config(start_state_time=2.0, heuristic=H_TSP_POINT_ALL if heur == "all" else H_TSP_POINT_K, tsp_k=k)
rib = np.asarray([[40 + 9 * i, 50 + 5 * (i % 3), 44 + 9 * i + 3 * (i % 2), 96 - 4 * (i % 4)] for i in range(nrib)], dtype=np.float64)
root = root_vertex(70.0, 30.0, 0.3, 2.5, 2.0, rib)
n = 700
sx, sy, sh = rng.uniform(20, 130, n), rng.uniform(20, 130, n), rng.uniform(0, 2 * np.pi, n)
outs = []
for threshold in ("0", "1000000000"):
    os.environ["PPGPU_PREPASS_MIN_EDGES"] = threshold
    ctx = api.Context(0)
    ctx.set_config(cfg); ctx.set_grid(None, 0.5); ctx.set_obstacles(None); ctx.set_vertices(root, rib); ctx.set_samples(sx, sy, sh)
    outs.append(_dense(torch, ctx, 1, n, 0xF, stride=10))
a, b = outs[0][0], outs[1][0]
bad = np.nonzero(a["h"] != b["h"])[0]
print("differing h:", len(bad), "of", len(a))
for e in bad[:10]:
    nr = (a["info"][e] >> 8) & 255
    print(e, "nrib", nr, "lanes", repr(a["h"][e]), "wave", repr(b["h"][e]), "end", a["end_x"][e], a["end_y"][e])
    print(outs[0][1][e][:nr])

import math
def dist(a, b): return math.sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]))
def tsp(point, ribs, sf, twoW, K, sortK):
    if not ribs: return sf
    if sortK:
        ribs = sorted(ribs, key=lambda r: -min(dist(point, r[0]), dist(point, r[1])))   # stable, descending
    best = float("inf")
    for i, r in enumerate(ribs):
        if i >= K: break
        rest = ribs[:i] + ribs[i + 1:]
        ln = dist(r[0], r[1])
        best = min(best, tsp(r[1], rest, max(sf + ln - twoW + dist(point, r[0]), 0.0), twoW, K, sortK))
        best = min(best, tsp(r[0], rest, max(sf + ln - twoW + dist(point, r[1]), 0.0), twoW, K, sortK))
    return best
for e in bad[:10]:
    nr = (a["info"][e] >> 8) & 255
    ribs = [((r[0], r[1]), (r[2], r[3])) for r in outs[0][1][e][:nr].tolist()]
    hd = tsp((float(a["end_x"][e]), float(a["end_y"][e])), ribs, 0.0, 2 * cfg.ribbon_width, 8 if heur == "all" else k, heur != "all")
    print(e, "python", repr(hd / cfg.max_speed * 1.0))
for e in bad[:10]:
    nr = (a["info"][e] >> 8) & 255
    g, fl = ctx.heuristic_host([[float(a["end_x"][e]), float(a["end_y"][e]), float(a["end_heading"][e])]], [outs[0][1][e][:nr].copy()])
    print(e, "pp_k_heuristic", repr(g[0]), "end pose of the wave run equal:", a["end_x"][e] == b["end_x"][e], a["end_y"][e] == b["end_y"][e],
          "children equal:", np.array_equal(outs[0][1][e], outs[1][1][e]))
cd = np.nonzero(np.any(outs[0][1].reshape(len(a), -1) != outs[1][1].reshape(len(a), -1), axis=1))[0]
print("edges whose child ribbons differ:", cd[:20], len(cd))
for e in cd[:4]:
    nr = (a["info"][e] >> 8) & 255
    print(e, "nrib", nr, (b["info"][e] >> 8) & 255, "steps", a["info"][e] >> 16, b["info"][e] >> 16, "flags", a["flags"][e], b["flags"][e])
    for i in range(nr):
        print("   ", [repr(float(x)) for x in outs[0][1][e][i]], "\n   ", [repr(float(x)) for x in outs[1][1][e][i]])
    print("   target", sx[e // 4], sy[e // 4], sh[e // 4], "cfg", e % 4)
