#!/usr/bin/env python3
"""Developer tool (GPU box): where the cover sweep's event loop spends its shader cycles — loading a window (time grid, 64 poses,
heading bits), continued runs, one-at-a-time events, run attempts after an event — from a -DPP_DBG_COUNTS -DPP_DBG_PHASES build
that returns the cycle counters in the record's param[] slots.  (Wall cycles of the wave: with four waves on a SIMD a part's
share includes the turns the other waves took.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpurun_out", "libppgpu_phases.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DPP_DBG_COUNTS", "-DPP_DBG_PHASES"] +
                      os.environ.get("RUN_COUNTS_FLAGS", "").split() + [
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"] = lib
if len(sys.argv) > 1 and sys.argv[1] == "approach":
    MODE = "approach"                       # the approach lanes' counters, in the records of the edges they finish themselves
else:
    MODE = "wave"
    os.environ["PPGPU_QUIET_FINISH"] = "0"      # the counters travel in record slots that only the wave writes
sys.path.insert(0, ROOT)
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE
w = workloads.config3(n_samples=8192)
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4 * n * 128, dtype=torch.uint8, device="cuda")
for _ in range(2):
    ctx.cost_edges_dense(0, 1, 0, n, 0xF, d.data_ptr()); ctx.synchronize()
r = d.cpu().numpy().view(RESULT_DTYPE)
p = r["param"]
if MODE == "approach":
    # quiet edges (finished by the lane) carry a negative value in this slot; a record the wave wrote has a cycle count there
    pro = r["coverage_completed_time"]; wloop, loop = p[:, 0] // 4294967296.0, p[:, 0] % 4294967296.0; fin = -1.0 - p[:, 1]; wev, ev = p[:, 2] // 1e6, p[:, 2] % 1e6
    q = p[:, 1] < 0
    print(f"edges {len(r)}, finished by their approach lane {q.mean():.3f}")
    print(f"  per wave (maximum over its lanes): before the loop {pro[q].mean():.0f} cycles, event loop {wloop[q].mean():.0f}, after it {fin[q].mean():.0f}; events of the busiest lane {wev[q].mean():.1f}")
    print(f"  per lane: event loop {loop[q].mean():.0f} cycles, events {ev[q].mean():.1f}  (cycles per event {loop[q].sum() / max(ev[q].sum(), 1):.0f})")
    for c in range(4):
        m = q.copy(); m[:] = False; m[c::4] = q[c::4]
        print(f"  cfg {c}: quiet {m.sum() / (len(r) / 4):.3f}  lane loop {loop[m].mean():8.0f}  events {ev[m].mean():6.1f}")
    sys.exit(0)
loop = r["coverage_completed_time"]
win, gen = p[:, 0] // 4294967296.0, p[:, 0] % 4294967296.0
cont, run = p[:, 1], p[:, 2]
tot = loop.sum()
other = loop - win - gen - cont - run
print(f"edges {len(r)}; event-loop cycles per edge {loop.mean():.0f} (edges with a loop: {(loop > 0).mean():.3f})")
for name, a in (("window load (time grid, 64 poses, heading bits)", win), ("continued runs", cont), ("one-at-a-time events", gen), ("run attempts after an event", run), ("rest of the loop", other)):
    print(f"  {name:50s} {a.sum() / tot:6.3f}   ({a.mean():8.0f} cycles per edge)")
for c in range(4):
    m = slice(c, None, 4)
    print(f"cfg {c}: loop {loop[m].mean():8.0f}  window {win[m].mean():8.0f}  continued {cont[m].mean():8.0f}  events {gen[m].mean():8.0f}  attempts {run[m].mean():8.0f}")
