import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE
w = workloads.config3()
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4*n*128, dtype=torch.uint8, device="cuda")
ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); ctx.synchronize()
r = d.cpu().numpy().view(RESULT_DTYPE)
steps = r['info']>>16
print("edges", len(r), "mean steps", steps.mean(), "chunks", np.ceil(steps/64).mean(), "max", steps.max())
print("infeasible", (r['flags']&1).mean(), "nobst", len(w.obst), "cfg", w.cfg)
for c in range(4):
    print(c, steps[c::4].mean())
print(np.percentile(steps,[10,25,50,75,90]))
nr = (r['info'] >> 8) & 255
print("child ribbon histogram:", np.bincount(nr), " throws:", int(((r['flags'] & 2) != 0).sum()))
