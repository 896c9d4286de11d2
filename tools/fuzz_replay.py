#!/usr/bin/env python3
"""GPU box: replay one world dumped by `FUZZ_DUMP=<round> tools/fuzz_parity.py ...` (gpurun_out/fuzz_case.pkl) with the
current library or a variant (PPGPU_LIB_OVERRIDE) and list the edges that differ from the oracle.
usage: tools/fuzz_replay.py [edge ...]"""
import os, pickle, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api
from path_planner_amd.types import RESULT_DTYPE, edge_pack, make_config
import oracle as orc
c = pickle.load(open(os.path.join(ROOT, "gpurun_out", "fuzz_case.pkl"), "rb"))
cfg = make_config(**c["cfg_kw"])
orc.O.ppo_set_ribbon_width(cfg.ribbon_width)
ctx = api.Context(0); ctx.set_config(cfg); ctx.set_grid(c["grid"], c["res"])
if c["model"] == "binary": ctx.set_obstacles(c["ob"]); world = orc.World(cfg, c["grid"], c["res"], c["ob"])
elif c["model"] == "gaussian": ctx.set_gaussian_obstacles(c["ob"]); world = orc.World(cfg, c["grid"], c["res"], gauss=c["ob"])
else: ctx.set_obstacles(None); world = orc.World(cfg, c["grid"], c["res"])
sx, sy, sh, rib, root = c["sx"], c["sy"], c["sh"], c["rib"], c["root"]
n = len(sx); ne = 4 * n
ctx.set_vertices(root, rib); ctx.set_samples(sx, sy, sh)
d_res = torch.zeros(ne * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
d_child = torch.zeros(ne * 20 * 4, dtype=torch.float64, device="cuda:0")
ctx.cost_edges_dense(0, 1, 0, n, 0xF, d_res.data_ptr(), d_child.data_ptr(), 20); ctx.synchronize()
gpu = d_res.cpu().numpy().view(RESULT_DTYPE)
e = edge_pack(np.zeros(ne, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
cpu, cchild = world.cost_edges(root, rib, sx, sy, sh, e, stride=20)
bad = np.nonzero((gpu["flags"] != cpu["flags"]) | (gpu["info"] != cpu["info"]))[0]
print(os.environ.get("PPGPU_LIB_OVERRIDE", "default"), "mismatching edges:", bad.tolist()[:20])
for b in [int(a) for a in sys.argv[1:]]:
    print(" edge", b, "gpu", hex(int(gpu["flags"][b])), (int(gpu["info"][b]) & 255, (int(gpu["info"][b]) >> 8) & 255, int(gpu["info"][b]) >> 16), "param", gpu["param"][b],
          "cpu", hex(int(cpu["flags"][b])), (int(cpu["info"][b]) & 255, (int(cpu["info"][b]) >> 8) & 255, int(cpu["info"][b]) >> 16))
gchild = d_child.cpu().numpy().reshape(ne, 20, 4)
cd = np.abs(gchild - cchild).max(axis=(1, 2))
worst = np.argsort(-cd)[:4]
print(" child ribbons: edges off by more than 1e-7 m:", np.nonzero(cd > 1e-7)[0].tolist()[:20])
for b in worst:
    if cd[b] <= 1e-7:
        break
    nr = (int(gpu["info"][b]) >> 8) & 255
    print(" edge", int(b), "cfg", int(b) % 4, "nrib", nr, "steps", int(gpu["info"][b]) >> 16, "flags", hex(int(gpu["flags"][b])), "max diff", cd[b])
    for i in range(nr):
        if np.abs(gchild[b, i] - cchild[b, i]).max() > 1e-7:
            print("    ribbon", i, "gpu", gchild[b, i].tolist(), "\n             cpu", cchild[b, i].tolist())
