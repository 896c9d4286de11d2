import faulthandler, sys, time, os
faulthandler.dump_traceback_later(60, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from path_planner_amd import api, workloads
from path_planner_amd.types import RESULT_DTYPE, H_TSP_POINT_ALL, edge_pack, make_config
from path_planner_amd.workloads import root_vertex
import oracle as orc
def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
rng = np.random.default_rng(11)
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = make_config(start_state_time=3.0, heuristic=H_TSP_POINT_ALL, tsp_k=2)
grid = np.zeros((300, 300), dtype=np.uint8); grid[40:60, 100:180] = 1; grid[200:230, 60:90] = 1
obst = workloads.obstacles(4, 9, 150.0, time=3.0, keep_free=(75, 75, 25))
rib = np.array([[60, 86 + 8 * i, 100 - 3 * i, 86 + 8 * i] for i in range(nr)], dtype=float)
root = root_vertex(75.0, 75.0, 0.4, 2.5, 3.0, rib)
n = 300
sx, sy, sh = rng.uniform(20, 130, n), rng.uniform(20, 130, n), rng.uniform(0, 2 * np.pi, n)
ctx = api.Context(0)
ctx.set_config(cfg); ctx.set_grid(grid, 0.5); ctx.set_obstacles(obst); ctx.set_vertices(root, rib); ctx.set_samples(sx, sy, sh)
ne = 4 * n
d_res = torch.zeros(ne * 128, dtype=torch.uint8, device="cuda:0")
d_child = torch.zeros(ne * 20 * 4, dtype=torch.float64, device="cuda:0")
log("launch")
ctx.cost_edges_dense(0, 1, 0, n, 0xF, d_res.data_ptr(), d_child.data_ptr(), 20)
ctx.synchronize()
log("gpu done")
gpu = d_res.cpu().numpy().view(RESULT_DTYPE)
log("nrib hist", np.unique((gpu["info"] >> 8) & 255, return_counts=True))
world = orc.World(cfg, grid, 0.5, obst)
e = edge_pack(np.zeros(ne, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
cpu = world.cost_edges(root, rib, sx, sy, sh, e)
log("cpu done", "max |dh|", np.nanmax(np.abs(gpu["h"] - cpu["h"])))
