#!/usr/bin/env python3
"""Developer tool (GPU box): distribution of coverage events per edge and class (every step inside a corridor counts as one event:
one-at-a-time events + steps absorbed by corridor / quiet runs), from a -DPP_DBG_COUNTS build; and what a lane-per-edge walk of
those events would cost a wavefront: the mean over groups of 64 consecutive edges of one class of the group's maximum."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpurun_out", "libppgpu_counts.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DPP_DBG_COUNTS",
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"] = lib
os.environ["PPGPU_QUIET_FINISH"] = "0"
sys.path.insert(0, ROOT)
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE
w = workloads.config3(n_samples=16384)
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4 * n * 128, dtype=torch.uint8, device="cuda")
ctx.cost_edges_dense(0, 1, 0, n, 0xF, d.data_ptr()); ctx.synchronize()
r = d.cpu().numpy().view(RESULT_DTYPE)
p = r["param"]
win, gen = p[:, 0] // 1e6, p[:, 0] % 1e6
cor, corl = p[:, 1] // 1e6, p[:, 1] % 1e6
qui, quil = p[:, 2] // 1e6, p[:, 2] % 1e6
ev = gen + corl + quil
ops = win + gen + cor + qui
steps = (r["info"] >> 16) & 0xffff
nrib = (r["info"] >> 8) & 0xff
for c in range(4):
    m = slice(c, None, 4)
    e, o = ev[m], ops[m]
    live = win[m] > 0
    el = e[live]
    print(f"cfg {c}: edges {e.size}, with a window {live.mean():.3f}; events per such edge mean {el.mean():.1f} p50 {np.percentile(el,50):.0f} p90 {np.percentile(el,90):.0f} "
          f"p99 {np.percentile(el,99):.0f} max {el.max():.0f}; wave ops per such edge {o[live].mean():.2f}; child ribbons mean {nrib[m][live].mean():.2f} max {nrib[m].max()}")
    g = el[: el.size // 64 * 64].reshape(-1, 64)
    print(f"        groups of 64 such edges: mean of max {g.max(axis=1).mean():.1f}, mean of mean {g.mean(axis=1).mean():.1f}")
    sg = np.sort(el)[: el.size // 64 * 64].reshape(-1, 64)
    print(f"        ... sorted by count first: mean of max {sg.max(axis=1).mean():.1f}")
    print(f"        one-at-a-time events mean {gen[m][live].mean():.2f} max {gen[m].max():.0f}")
