"""Developer tool (GPU box): HIP-event times of the costing launch's kernel groups for each of the four edge configurations
of the bench workload alone (dense launch with cfg_mask 1, 2, 4, 8) and for all four: which class of edges the time goes to.
PPGPU_LIB_OVERRIDE selects a variant library."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_planner_amd import api, workloads
w = workloads.config3()
ctx = api.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4 * n * 128, dtype=torch.uint8, device="cuda")
ctx.enable_timing(True)
print(os.environ.get("PPGPU_LIB_OVERRIDE", "default"))
for mask in (1, 2, 4, 8, 0xF):
    ts = []
    for i in range(5):
        ctx.cost_edges_dense(0, 1, 0, n, mask, d.data_ptr()); ts.append(ctx.last_timing())
    print("mask %2d" % mask, " solve/pose/cover/heur ms:", np.round(np.min(np.array(ts[1:]), axis=0), 3), " edges the cover sweep visited:", ctx.last_cover_edges(), flush=True)
