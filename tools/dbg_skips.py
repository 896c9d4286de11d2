import os, subprocess, sys
ROOT="/root/repo"
lib=os.path.join(ROOT,"gpurun_out","libppgpu_dbgskips.so")
subprocess.check_call(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-ffp-contract=off","-fPIC","-shared","-std=c++17","-DPP_DBG_SKIPS",os.path.join(ROOT,"path_planner_amd","csrc","ppgpu.hip"),"-o",lib,"-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"]=lib
sys.path.insert(0,ROOT)
import numpy as np, torch
from path_planner_amd import api, workloads
w=workloads.config3()
ctx=api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid,w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(),w.ribbons4)
ctx.sampler_init(w.bounds6,w.seed,w.ribbons4); n=ctx.sampler_add(w.n_samples)
d=torch.zeros(4*n*128,dtype=torch.uint8,device="cuda")
ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); ctx.synchronize()
for mask in (1,2,4,8):
    ctx.cost_edges_dense(0,1,0,n,mask,d.data_ptr()); ctx.synchronize()
ctx.set_obstacles(None)
ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); ctx.synchronize()
