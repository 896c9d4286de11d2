import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from path_planner_amd import api, workloads
w = workloads.config3()
ctx = api.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4*n*128, dtype=torch.uint8, device="cuda")
ctx.enable_timing(True)
ts=[]
for i in range(5):
    ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); ts.append(ctx.last_timing())
print(os.environ.get("PPGPU_LIB_OVERRIDE","default"), " solve/pose/cover/heur ms:", np.round(np.min(np.array(ts[1:]),axis=0),3))
