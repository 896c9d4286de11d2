#!/usr/bin/env python3
"""Developer tool (GPU box): does a device-to-host copy of one launch's records overlap with the next costing launch?  Times the launch
(wall, and the library's HIP events between its kernels) alone, beside an SDMA copy (ppgpu_copy_engine_read) and beside torch's
non-blocking copy (a blit kernel on a second stream)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_planner_amd import api, workloads
w = workloads.config3()
ctx = api.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
nb = 4 * n * 128
d = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
h = torch.empty(nb, dtype=torch.uint8, pin_memory=True)
ctx.enable_timing(True)
cs = torch.cuda.Stream()
def launch(mode):
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "sdma":
        ctx.copy_engine_read(h.data_ptr(), d[1].data_ptr(), nb)
    elif mode == "blit":
        with torch.cuda.stream(cs):
            h.copy_(d[1], non_blocking=True)
    t1 = time.perf_counter()
    ctx.cost_edges_dense(0, 1, 0, n, 0xF, d[0].data_ptr())
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    if mode == "sdma":
        ctx.copy_engine_wait()
    elif mode == "blit":
        cs.synchronize()
    t4 = time.perf_counter()
    return [1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0), 1e3 * (t4 - t0)] + list(ctx.last_timing())
for mode in ("alone", "sdma", "blit", "alone", "sdma", "blit"):
    r = np.array([launch(mode) for _ in range(6)][1:])
    m = np.median(r, axis=0)
    print(f"{mode:6s} issue copy {m[0]:6.3f} ms | launch calls {m[1]:6.3f} | kernels done at {m[2]:6.3f} | all done at {m[3]:6.3f} | solve/pose/cover/heur {np.round(m[4:], 3)}", flush=True)
