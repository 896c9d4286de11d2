"""Deterministic synthetic inputs for the BASELINE.json configs (SURVEY.md section 8 d).

Everything derives from splitmix64 streams with fixed seeds, so the same bytes are produced
in this container, on the GPU box and in later rounds.  No file or network input.
"""
import math

import numpy as np

from .types import VERTEX_DTYPE, make_config, H_TSP_POINT_K, H_MAX_DISTANCE

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, a=0.0, b=1.0):
        return a + (b - a) * ((self.next() >> 11) * (1.0 / (1 << 53)))


def blob_grid(n, res, frac, seed, start_xy, keep_free_radius=10.0, blob=32):
    """n x n byte grid (row 0 = y in [0,res)), `frac` of the area blocked by blob x blob squares placed
    by splitmix64(seed) until coverage >= frac, keeping a disc around the start free."""
    g = np.zeros((n, n), dtype=np.uint8)
    if frac <= 0:
        return g
    rng = SplitMix64(seed)
    target = int(math.ceil(frac * n * n))
    covered = 0
    sx, sy = start_xy
    while covered < target:
        c0 = rng.next() % (n - blob + 1)
        r0 = rng.next() % (n - blob + 1)
        # closest point of the square to the start
        x0, x1 = c0 * res, (c0 + blob) * res
        y0, y1 = r0 * res, (r0 + blob) * res
        dx = max(x0 - sx, 0.0, sx - x1)
        dy = max(y0 - sy, 0.0, sy - y1)
        if math.hypot(dx, dy) < keep_free_radius:
            continue
        sub = g[r0:r0 + blob, c0:c0 + blob]
        covered += int(sub.size - np.count_nonzero(sub))
        sub[:] = 1
    return g


def obstacles(n, seed, extent, time=1.0, width=10.0, length=30.0, max_speed=3.0, keep_free=None):
    """n rows {x, y, heading, speed, time, width, length} (BinaryDynamicObstaclesManager::update arguments).
    keep_free = (x, y, radius): redraw an obstacle whose centre starts closer than radius to (x, y), so that the
    vehicle does not begin the plan inside a box."""
    rng = SplitMix64(seed)
    o = np.zeros((n, 7), dtype=np.float64)
    i = 0
    while i < n:
        row = [rng.uniform(0, extent), rng.uniform(0, extent), rng.uniform(0, 2 * math.pi), rng.uniform(0, max_speed),
               time, width, length]
        if keep_free is not None and math.hypot(row[0] - keep_free[0], row[1] - keep_free[1]) < keep_free[2]:
            continue
        o[i] = row
        i += 1
    return o


def root_vertex(x, y, heading, speed, time, ribbons4, cct=-1.0):
    v = np.zeros(1, dtype=VERTEX_DTYPE)
    v["x"], v["y"], v["heading"], v["speed"], v["time"] = x, y, heading, speed, time
    v["g"] = 0.0
    v["coverage_completed_time"] = cct
    v["ribbon_offset"] = 0
    v["ribbon_count"] = len(ribbons4)
    return v


class Workload:
    """One BASELINE config: world + config + sampler parameters."""

    def __init__(self, name, grid, res, obst, ribbons4, start5, n_samples, seed, cfg, n_vertices=1):
        self.name, self.grid, self.res, self.obst = name, grid, res, obst
        self.ribbons4 = np.asarray(ribbons4, dtype=np.float64).reshape(-1, 4)
        self.start5 = np.asarray(start5, dtype=np.float64)
        self.n_samples, self.seed, self.cfg, self.n_vertices = n_samples, seed, cfg, n_vertices

    @property
    def bounds6(self):
        """AStarPlanner::plan's sampling box (AStarPlanner.cpp:27-32)."""
        mag = self.cfg.max_speed * self.cfg.time_horizon
        x, y = self.start5[0], self.start5[1]
        if self.grid is None:
            ext = [-1.7976931348623157e308, 1.7976931348623157e308, -1.7976931348623157e308, 1.7976931348623157e308]
        else:
            ext = [0.0, self.grid.shape[1] * self.res, 0.0, self.grid.shape[0] * self.res]
        return np.array([max(x - mag, ext[0]), min(x + mag, ext[1]), max(y - mag, ext[2]), min(y + mag, ext[3]),
                         self.cfg.max_speed, self.cfg.max_speed], dtype=np.float64)

    def root(self):
        s = self.start5
        return root_vertex(s[0], s[1], s[2], self.cfg.max_speed, s[4], self.ribbons4)


def config1():
    """256x256 empty grid, res 1.0, 64 samples, 1 ribbon (CPU-runnable plumbing case)."""
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    grid = np.zeros((256, 256), dtype=np.uint8)
    return Workload("cfg1_256_empty_64", grid, 1.0, None, [[118, 138, 138, 138]], [128, 128, 0, 2.5, 1], 64, 7, cfg)


def config2(n_samples=4096):
    """1024x1024 grid res 0.2 m, 10 % blocked (32-cell squares, seed 1), no dynamic obstacles, 2 ribbons."""
    cfg = make_config(start_state_time=1.0, heuristic=H_TSP_POINT_K, tsp_k=2)
    c = 1024 * 0.2 / 2
    grid = blob_grid(1024, 0.2, 0.10, 1, (c, c))
    rib = [[c - 20, c + 10, c + 20, c + 10], [c - 20, c + 30, c + 20, c + 30]]
    return Workload("cfg2_1024_10pct_4096", grid, 0.2, None, rib, [c, c, 0, 2.5, 1], n_samples, 7, cfg)


def config3(n_samples=65536, n_obst=16, keep_free=None):
    """2048x2048 grid res 0.1 m, 10 % blocked (seed 2), 16 moving boxes (seed 3) with positions uniform in the map (SURVEY 8 d;
    keep_free = (x, y, radius) redraws boxes that start closer than radius to a point), 5 ribbons, TSP K=2 heuristic."""
    cfg = make_config(start_state_time=1.0, heuristic=H_TSP_POINT_K, tsp_k=2)
    ext = 2048 * 0.1
    c = ext / 2
    grid = blob_grid(2048, 0.1, 0.10, 2, (c, c))
    obst = obstacles(n_obst, 3, ext, keep_free=keep_free) if n_obst else None
    rib = [[c - 20, c + 12 + 8 * i, c + 20, c + 12 + 8 * i] for i in range(5)]
    return Workload("cfg3_2048_10pct_65536_obst%d" % n_obst, grid, 0.1, obst, rib, [c, c, 0, 2.5, 1], n_samples, 7, cfg,
                    n_vertices=64)


def by_name(name):
    return {"cfg1": config1, "cfg2": config2, "cfg3": config3}[name]()
