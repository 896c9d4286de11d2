#!/usr/bin/env python3
"""Developer tool (GPU box): what the cover sweep's wavefront does on one edge of the bench workload, window by window
(-DPP_DBG_TRACE=<record index> build).  usage: tools/trace_edge.py <record index> [n_samples]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
eg = int(sys.argv[1]); ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
os.makedirs(os.path.join(ROOT, "gpurun_out", "abl"), exist_ok=True)
lib = os.path.join(ROOT, "gpurun_out", "abl", "libppgpu_trace.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", f"-DPP_DBG_TRACE={eg}",
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
os.environ["PPGPU_LIB_OVERRIDE"] = lib
sys.path.insert(0, ROOT)
import numpy as np, torch
from path_planner_amd import api, workloads
w = workloads.config3(n_samples=ns)
ctx = api.Context(0)
ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst); ctx.set_vertices(w.root(), w.ribbons4)
ctx.sampler_init(w.bounds6, w.seed, w.ribbons4); n = ctx.sampler_add(w.n_samples)
d = torch.zeros(4 * n * 128, dtype=torch.uint8, device="cuda")
ctx.cost_edges_dense(0, 1, 0, n, 0xF, d.data_ptr()); ctx.synchronize()
