#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE's own objects (oracle/_ref/libpp_ref.so, built by
`make -C oracle ref` from the sources under /root/reference: State.cpp, Ribbon.cpp, Map.cpp,
GridWorldMap.cpp, BinaryDynamicObstaclesManager.cpp).  Only inputs and the reference's outputs are
stored (doubles as C99 hex strings, so they round-trip bit-exactly); no reference source text.

Run in the build container only (the reference tree does not travel):  python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402

REF = orc.REF
assert REF is not None, "build oracle/_ref first: make -C oracle ref"


def hx(v):
    return [float(x).hex() for x in np.asarray(v, dtype=np.float64).ravel()]


def arr(n):
    return np.zeros(n, dtype=np.float64)


def ribbon_cases(rng, n):
    cases = []
    for i in range(n):
        kind = i % 4
        r = rng.uniform(-100, 100, 4)
        if kind == 1:   # axis-aligned
            r[3] = r[1]
        if kind == 2:   # short
            r[2:] = r[:2] + rng.uniform(-3, 3, 2)
        # point: near the line half of the time
        t = rng.uniform(-0.2, 1.2)
        base = r[:2] + t * (r[2:] - r[:2])
        off = rng.uniform(-3, 3, 2) if i % 2 else rng.uniform(-40, 40, 2)
        p = base + off
        w = [1.5, 2.0, 0.5][i % 3]
        REF.ref_set_ribbon_width(w)
        rr = np.ascontiguousarray(r)
        proj = arr(2)
        REF.ref_ribbon_projection(rr.ctypes.data, p[0], p[1], proj.ctypes.data)
        c = dict(w=float(w).hex(), r=hx(r), p=hx(p), proj=hx(proj))
        c["contains_projection"] = REF.ref_ribbon_contains_projection(rr.ctypes.data, proj[0], proj[1])
        c["contains"] = [REF.ref_ribbon_contains(rr.ctypes.data, p[0], p[1], 0), REF.ref_ribbon_contains(rr.ctypes.data, p[0], p[1], 1)]
        c["distance"] = float(REF.ref_ribbon_distance(rr.ctypes.data, p[0], p[1])).hex()
        c["length"] = float(REF.ref_ribbon_length(rr.ctypes.data)).hex()
        c["covered"] = [REF.ref_ribbon_covered(rr.ctypes.data, 0), REF.ref_ribbon_covered(rr.ctypes.data, 1)]
        for strict in (0, 1):
            r2 = rr.copy()
            front = arr(4)
            REF.ref_ribbon_split(r2.ctypes.data, p[0], p[1], strict, front.ctypes.data)
            c["split%d" % strict] = dict(rest=hx(r2), front=hx(front))
        s5, e5, pj = arr(5), arr(5), arr(5)
        REF.ref_ribbon_end_states(rr.ctypes.data, s5.ctypes.data, e5.ctypes.data)
        REF.ref_ribbon_projection_as_state(rr.ctypes.data, p[0], p[1], pj.ctypes.data)
        c["start_state"], c["end_state"], c["projection_state"] = hx(s5), hx(e5), hx(pj)
        cases.append(c)
    REF.ref_set_ribbon_width(1.5)
    return cases


MAP_TEXT = """0.5
....#......#....
...##...........
................
#..............#
......####......
......####......
................
.#..........#...
................
....#....#......
"""

RAGGED_MAP_TEXT = """2
..#.....
.#...
......##
#...
"""


def grid_cases(rng, text, n):
    with tempfile.NamedTemporaryFile("w", suffix=".map", delete=False) as f:
        f.write(text)
        path = f.name
    g = REF.ref_grid_load(path.encode())
    ext = arr(4)
    REF.ref_grid_extremes(g, ext.ctypes.data)
    res = REF.ref_grid_resolution(g)
    xs = rng.uniform(ext[0] - 1, ext[1] + 1, n)
    ys = rng.uniform(ext[2] - 1, ext[3] + 1, n)
    # exact cell boundaries and the map edges
    k = np.arange(0, 12)
    xs = np.concatenate([xs, k * res, [ext[1], ext[1] - 1e-12, 0.0, -0.0, -1e-300]])
    ys = np.concatenate([ys, k * res * 0.5, [ext[3], ext[3] - 1e-12, 0.0, -0.0, 0.25]])
    m = min(len(xs), len(ys))
    xs, ys = np.ascontiguousarray(xs[:m]), np.ascontiguousarray(ys[:m])
    out = np.zeros(m, dtype=np.uint8)
    REF.ref_grid_is_blocked_many(g, m, xs.ctypes.data, ys.ctypes.data, out.ctypes.data)
    REF.ref_grid_free(g)
    os.unlink(path)
    return dict(text=text, extremes=hx(ext), resolution=float(res).hex(), x=hx(xs), y=hx(ys), blocked=[int(v) for v in out])


def obstacle_cases(rng, n_obst, n_q):
    m = REF.ref_obst_create()
    obst = []
    for i in range(n_obst):
        o = [rng.uniform(0, 200), rng.uniform(0, 200), rng.uniform(0, 2 * np.pi), rng.uniform(0, 3), rng.uniform(0, 5),
             rng.uniform(2, 12), rng.uniform(5, 30)]
        obst.append(o)
        REF.ref_obst_update(m, i + 1, *o)
    qs = []
    for i in range(n_q):
        o = obst[i % n_obst]
        t = rng.uniform(0, 40)
        # aim near the obstacle's projected position half of the time
        yaw = np.pi / 2 - o[2]
        cx, cy = o[0] + o[3] * (t - o[4]) * np.cos(yaw), o[1] + o[3] * (t - o[4]) * np.sin(yaw)
        if i % 2:
            x, y = cx + rng.uniform(-20, 20), cy + rng.uniform(-20, 20)
        else:
            x, y = rng.uniform(0, 200), rng.uniform(0, 200)
        qs.append(dict(q=hx([x, y, t]), loose=float(REF.ref_obst_collision_exists(m, x, y, t, 0)),
                       strict=float(REF.ref_obst_collision_exists(m, x, y, t, 1))))
    REF.ref_obst_free(m)
    return dict(obstacles=[hx(o) for o in obst], queries=qs)


def state_cases(rng, n):
    cases = []
    for i in range(n):
        s = np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(0, 2 * np.pi), rng.uniform(0, 3), rng.uniform(0, 100)])
        if i % 7 == 0:
            s[2] = [0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 7.0, -0.3][(i // 7) % 7]
        tgt = rng.uniform(-100, 100, 2)
        d = rng.uniform(-5, 5)
        mv = s.copy()
        REF.ref_state_move(mv.ctypes.data, d)
        ps = arr(5)
        REF.ref_state_push(s.ctypes.data, d, ps.ctypes.data)
        cases.append(dict(s=hx(s), target=hx(tgt), d=float(d).hex(),
                          yaw=float(REF.ref_state_yaw(s[2])).hex(),
                          heading_to=float(REF.ref_state_heading_to(s[0], s[1], tgt[0], tgt[1])).hex(),
                          distance_to=float(REF.ref_state_distance_to(s.ctypes.data, tgt[0], tgt[1])).hex(),
                          moved=hx(mv), pushed=hx(ps)))
    return cases


def main():
    rng = np.random.default_rng(20261003)
    out = {
        "ribbon_ops.json": ribbon_cases(rng, 400),
        "grid_map.json": [grid_cases(rng, MAP_TEXT, 300), grid_cases(rng, RAGGED_MAP_TEXT, 200)],
        "binary_obstacles.json": obstacle_cases(rng, 6, 400),
        "state_ops.json": state_cases(rng, 200),
    }
    bm = arr(4)
    REF.ref_base_map_extremes(bm.ctypes.data)
    out["base_map.json"] = dict(extremes=hx(bm), blocked=[REF.ref_base_map_is_blocked(0.0, 0.0), REF.ref_base_map_is_blocked(-1e9, 1e9)])
    for name, data in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
