import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpurun_out", "libppgpu_dbg.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DPP_DBG_EVENTS",
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"] = lib
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE, edge_pack, VERTEX_DTYPE
import oracle as orc, ctypes as C
w = workloads.config3(n_samples=2048)
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4*n*128, dtype=torch.uint8, device="cuda")
ctx.cost_edges_dense(0,1,0,n,0xF,d.data_ptr()); ctx.synchronize()
g = d.cpu().numpy().view(RESULT_DTYPE)
gen = (g["info"] >> 16).astype(np.int64)
world = orc.World(w.cfg, w.grid, w.res, w.obst)
cs = world.add_samples(w.bounds6, w.seed, w.ribbons4, 0, w.n_samples)
e = edge_pack(np.zeros(4*n,dtype=np.uint64), np.repeat(np.arange(n),4), np.tile(np.arange(4),n))
st = np.zeros((4*n,2),dtype=np.int32)
v = np.ascontiguousarray(w.root(), dtype=VERTEX_DTYPE); r = np.ascontiguousarray(w.ribbons4)
sx,sy,sh = [np.ascontiguousarray(cs[:,i]) for i in range(3)]
orc.O.ppo_edge_event_stats(world.h, v.ctypes.data, r.ctypes.data, sx.ctypes.data, sy.ctypes.data, sh.ctypes.data, 4*n, e.ctypes.data, st.ctypes.data)
print("oracle events mean", st[:,0].mean(), "mutations", st[:,1].mean(), " gpu generic events mean", gen.mean())
for c in range(4): print("cfg", c, "oracle events", st[c::4,0].mean(), "gpu generic", gen[c::4].mean())
