import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from path_planner_amd import workloads
from path_planner_amd.types import F_THROWS, F_INFEASIBLE
from test_gpu_parity import _setup, _dense
w = workloads.config3(n_samples=16384)
ctx, world, n, cs = _setup(w, 16384)
gpu, gchild = _dense(torch, ctx, 1, n, 0xF)
rib = np.asarray(w.ribbons4).reshape(-1, 4)
nr = (gpu["info"] >> 8) & 255
live = (gpu["flags"] & F_THROWS) == 0
same = live & (nr == len(rib)) & np.all(gchild[:, :len(rib)].reshape(len(gpu), -1) == rib.reshape(-1), axis=1)
print("edges", len(gpu), "unchanged ribbon list", same.mean(), "infeasible", ((gpu["flags"] & F_INFEASIBLE) != 0).mean())
for c in range(4):
    print("cfg", c, "unchanged", same[c::4].mean())
