#!/usr/bin/env python3
"""GPU box: how often the device agrees with the oracle (glibc) to the last bit — Dubins parameters, approximate cost, end pose — on
config 3's first 4 096 samples from the root and from 12 of its children.  usage: tools/bit_agreement.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE, VERTEX_DTYPE, edge_pack, F_THROWS, F_INFEASIBLE
import oracle as orc
from test_gpu_parity import _setup, _dense

w = workloads.config3(n_samples=4096)
ctx, world, n, cs = _setup(w, 4096)
gpu, gchild = _dense(torch, ctx, 1, n, 0xF)
e = edge_pack(np.zeros(4 * n, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
cpu, cchild = world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e, stride=8)


def report(tag, g, c):
    live = ((c["flags"] & F_THROWS) == 0) & ((g["info"] & 255) == (c["info"] & 255))
    out = {"edges": int(live.sum())}
    out["params"] = float(np.mean(np.all(g["param"][live] == c["param"][live], axis=1)))
    for f in ("approx_cost", "end_x", "end_y", "end_heading", "end_time", "g"):
        out[f] = float(np.mean(g[f][live] == c[f][live]))
    print(tag, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in out.items()}, flush=True)


report("root   ", gpu, cpu)
feas = np.nonzero(((cpu["flags"] & (F_THROWS | F_INFEASIBLE)) == 0))[0]
pick = feas[:: max(1, len(feas) // 12)][:12]
v = np.zeros(len(pick), dtype=VERTEX_DTYPE)
pool, off = [], 0
for k2, ei in enumerate(pick):
    r_ = cpu[ei]; nr = int((r_["info"] >> 8) & 0xFF)
    v[k2] = (r_["end_x"], r_["end_y"], r_["end_heading"], r_["end_speed"], r_["end_time"], r_["g"], r_["coverage_completed_time"], off, nr)
    pool.append(cchild[ei, :nr]); off += nr
pool = np.concatenate(pool)
ctx.set_vertices(v, pool)
m = 1024
vi = np.repeat(np.arange(len(pick)), m * 4); ti = np.tile(np.repeat(np.arange(m), 4), len(pick)); ci = np.tile(np.arange(4), len(pick) * m)
e2 = edge_pack(vi.astype(np.uint64), ti, ci)
d_e = torch.from_numpy(e2.view(np.int64)).to("cuda:0")
d_res = torch.zeros(len(e2) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
d_child = torch.zeros(len(e2) * 16 * 4, dtype=torch.float64, device="cuda:0")
ctx.cost_edges_list(len(e2), d_e.data_ptr(), d_res.data_ptr(), d_child.data_ptr(), 16); ctx.synchronize()
g2 = d_res.cpu().numpy().view(RESULT_DTYPE)
c2, _ = world.cost_edges(v, pool, cs[:, 0], cs[:, 1], cs[:, 2], e2, stride=16, threads=8)
report("children", g2, c2)
