#!/usr/bin/env python3
"""Developer tool (GPU box): per-edge counts of what the cover sweep does (windows loaded, one-at-a-time events, corridor /
quiet run attempts and the steps they absorbed), from a -DPP_DBG_COUNTS build that returns them in the param[] slots."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpurun_out", "libppgpu_counts.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DPP_DBG_COUNTS"] +
                      os.environ.get("RUN_COUNTS_FLAGS", "").split() + [
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"] = lib
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
ctx.cost_edges_dense(0, 1, 0, n, 0xF, d.data_ptr()); ctx.synchronize()
r = d.cpu().numpy().view(RESULT_DTYPE)
p = r["param"]
win, gen = p[:, 0] // 1e6, p[:, 0] % 1e6
cor, corl = p[:, 1] // 1e6, p[:, 1] % 1e6
qui, quil = p[:, 2] // 1e6, p[:, 2] % 1e6
cc = r["coverage_completed_time"]
restfar, far, noch, inpl = cc // 1e9, (cc % 1e9) // 1e6, (cc % 1e6) // 1e3, cc % 1e3
for c in range(4):
    m = slice(c, None, 4)
    print(f"cfg {c}: windows {win[m].mean():6.2f}  generic events {gen[m].mean():6.2f}  corridor runs {cor[m].mean():6.2f} (steps {corl[m].mean():7.1f})"
          f"  quiet runs {qui[m].mean():6.2f} (steps {quil[m].mean():7.1f})")
print(f"generic events by kind: far fast path {far.mean():.2f}, near but nothing changed {noch.mean():.2f}, in-place endpoint move {inpl.mean():.2f}, "
      f"other (split / erase / compaction) {(gen - far - noch - inpl).mean():.2f}")
print(f"edges whose event loop ended on 'the rest of the curve is far': {restfar.mean():.3f}; far events of the others: {far[restfar == 0].mean():.2f}")
print(f"all  : windows {win.mean():6.2f}  generic events {gen.mean():6.2f}  corridor runs {cor.mean():6.2f} (steps {corl.mean():7.1f})"
      f"  quiet runs {qui.mean():6.2f} (steps {quil.mean():7.1f})")
