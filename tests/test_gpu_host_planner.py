"""-m gpu: the C++ host planner (path_planner_amd/host, GpuAStarPlanner::plan through the C ABI) against the CPU
oracle's restatement of AStarPlanner::plan, both driven by the same injected clock so that the seed and the number of
clock polls — hence iterations and expansions — are reproducible (PlannerConfig::setNowFunction, PlannerConfig.h:110-114).

Bar (BASELINE.json north_star): identical first-goal iteration index; trajectory costs within 1e-5 relative."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "path_planner_amd", "host", "plan_cli")


def _write_map(grid, res, path):
    with open(path, "w") as f:
        f.write(repr(float(res)) + "\n")
        for row in grid[::-1]:          # last text line is y = 0 (GridWorldMap.cpp:25)
            f.write("".join("#" if c else "." for c in row) + "\n")


def _scenario(w, path, map_path, t0, dt, budget_calls, initial_samples, prev=None, start=None, gauss=None, speculation=None,
              brown=False, devices=None):
    c = w.cfg
    lines = []
    if speculation is not None:
        lines.append(f"cfg speculation {speculation}")
    if brown:
        lines.append("cfg use_brown_paths 1")
    if devices is not None:
        lines.append("devices " + " ".join(str(d) for d in devices))
    for k in ("max_speed", "slow_speed", "turning_radius", "coverage_turning_radius", "time_horizon", "time_minimum",
              "collision_checking_increment", "branching_factor"):
        lines.append(f"cfg {k} {getattr(c, k)!r}")
    lines.append(f"cfg initial_samples {initial_samples}")
    s = w.start5 if start is None else start
    lines.append("start " + " ".join(repr(float(v)) for v in s))
    lines.append(f"heuristic {c.heuristic} {c.tsp_k} {c.heuristic_turning_radius!r}")
    lines.append(f"ribbon_width {c.ribbon_width!r}")
    for r in w.ribbons4:
        lines.append("ribbon " + " ".join(repr(float(v)) for v in r))
    if gauss is not None:
        for o in gauss:
            lines.append("gaussian " + " ".join(repr(float(v)) for v in o))
    elif w.obst is not None:
        for o in w.obst:
            lines.append("obstacle " + " ".join(repr(float(v)) for v in o))
    if map_path:
        lines.append(f"map_file {map_path}")
    lines.append(f"clock {t0!r} {dt!r}")
    lines.append(f"time_remaining {budget_calls * dt!r}")
    if prev is not None:
        for p in prev:
            lines.append("prev " + " ".join(repr(float(v)) if i != 7 else str(int(v)) for i, v in enumerate(p)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _run_cli(path):
    assert os.path.exists(CLI), "build the host library: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.run([CLI, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def _same_curve(a, b, n=17):
    """Two plan segments {qi[3], param[3], rho, type, speed, start, end} trace the same poses (1e-4 m, 1e-4 rad)."""
    import oracle as orc
    for u in np.linspace(0.0, 1.0, n):
        pa = orc.dubins_sample(a[:8], u * (a[10] - a[9]) * a[8] * (1 - 1e-9))
        pb = orc.dubins_sample(b[:8], u * (b[10] - b[9]) * b[8] * (1 - 1e-9))
        if pa[0] != 0 or pb[0] != 0:
            return False
        d = pa[1] - pb[1]
        d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
        if np.max(np.abs(d)) > 1e-4:
            return False
    return True


def _compare(host, st, plan, allow_order_fallbacks=False):
    assert "exception" not in host, host
    assert host["samples"] == st.samples
    assert host["iterations"] == st.iterations
    assert host["expanded"] == st.expanded
    assert host["generated"] == st.generated
    assert host["first_goal_iteration"] == st.first_goal_iteration       # identical first-goal iteration index
    assert host["plan_depth"] == st.plan_depth
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1.0)
    assert rel(host["plan_f"], st.plan_f) <= 1e-5                           # trajectory cost within 1e-5 relative
    assert rel(host["plan_h"], st.plan_h) <= 1e-5
    assert rel(host["plan_time_penalty"], st.plan_time_penalty) <= 1e-5
    assert rel(host["plan_collision_penalty"], st.plan_collision_penalty) <= 1e-5      # an integer multiple of 600 for the binary model
    hp = np.array(host["plan"], dtype=np.float64).reshape(-1, 11)
    assert hp.shape == plan.shape
    if len(hp):
        # same start, same end time, contiguous in time
        assert np.allclose(hp[0, [0, 1, 2, 9]], plan[0, [0, 1, 2, 9]], rtol=1e-9, atol=1e-9)
        assert abs(hp[-1, 10] - plan[-1, 10]) <= 1e-5 * max(1.0, abs(plan[-1, 10]))
        assert np.all(np.abs(hp[1:, 9] - hp[:-1, 10]) < 1e-9)
        # segment by segment: same Dubins word, parameters / times / speed within 1e-5 — including among children of EXACTLY
        # equal f, which std::pop_heap surfaces in an order that depends on the order expand() pushed them in (the device
        # replays the reference's heap-array order: ppgpu_expand_order).  One thing may differ in name only: a curve with a
        # zero-length arc is the same curve under two words (RSL / RSR with no final turn ...), their lengths tie exactly, and
        # which of them `cost < best` keeps hangs on the last bit of libm's atan2; such a segment is
        # compared as geometry.
        # The same goes for a straight line (both arcs of length zero) solved at the two turning radii of a vertex's edge
        # configurations: one trajectory, two descriptions of exactly equal cost; which of the two equal-f children is popped first
        # then hangs on an ulp of h (the child ribbons carry corridor-run rounding).
        for a, b in zip(hp, plan):
            if a[7] == b[7] and np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) <= 1e-5:
                continue
            assert min(a[3], a[5]) <= 1e-9 and min(b[3], b[5]) <= 1e-9, (hp, plan)
            assert np.max(np.abs(a[8:] - b[8:]) / np.maximum(np.abs(b[8:]), 1.0)) <= 1e-5, (hp, plan)
            assert _same_curve(a, b), (hp, plan)
    # lists whose push order the device could not replay (more candidates than it sorts: thousands of samples at EQUAL distance, e.g.
    # ribbon projections piled up on a nearly covered ribbon, where the reference's own order is that of its heap sort's ties)
    assert allow_order_fallbacks or host.get("order_fallbacks", 0) == 0


@pytest.mark.parametrize("name,init,calls", [("cfg1", 64, 60), ("cfg2", 256, 40), ("cfg3", 512, 30)])
def test_host_planner_matches_oracle_plan(name, init, calls, prepass_route):
    import oracle as orc
    from path_planner_amd import workloads
    w = workloads.by_name(name)
    orc.O.ppo_set_ribbon_width(w.cfg.ribbon_width)
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    t0, dt = 1000.0, 1e-3
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, t0, dt, calls, init)
        host = _run_cli(sc)
        rc, st, plan, itf, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init)
        assert rc == 0
        print(name, {k: host[k] for k in host if k != "plan"})
        assert st.first_goal_iteration >= 0 and st.expanded >= 3
        _compare(host, st, plan)

        # second cycle: move one second along the plan and replan with the previous plan as a seed (AStarPlanner.cpp:46-59)
        if len(plan):
            seg = plan[0]
            s1 = np.array([seg[0], seg[1], 0.0, seg[8], seg[9]])
            # sample the first segment one second in, with the oracle's wrapper
            out = np.zeros(3)
            e, q = orc.dubins_sample(seg[:8], min(1.0 * seg[8], (seg[10] - seg[9]) * seg[8]))
            assert e == 0
            import math
            hdg = math.pi / 2 - q[2]
            if hdg < 0:
                hdg += 2 * math.pi
            start2 = np.array([q[0], q[1], hdg, seg[8], seg[9] + 1.0])
            sc2 = os.path.join(d, "s2.txt")
            _scenario(w, sc2, mp, t0 + 1.0, dt, calls, init, prev=plan, start=start2)
            host2 = _run_cli(sc2)
            w.cfg.start_state_time = float(start2[4])
            world.set_config(w.cfg)
            rc2, st2, plan2, _, _ = world.plan(w.ribbons4, start2, calls * dt, t0 + 1.0, dt, initial_samples=init, prev11=plan)
            assert rc2 == 0
            print(name, "replan", {k: host2[k] for k in host2 if k != "plan"})
            _compare(host2, st2, plan2)


def test_host_planner_with_gaussian_obstacles():
    """The same plan() through GaussianDynamicObstaclesManager (unordered_map of mmsi -> obstacle on the host, uploaded with
    ppgpu_set_gaussian_obstacles): collision penalties are sums of densities, so a plan that passes near an obstacle
    carries a non-integer penalty that must agree with the oracle's planner to 1e-5."""
    import oracle as orc
    from path_planner_amd import workloads
    w = workloads.by_name("cfg2")
    orc.O.ppo_set_ribbon_width(w.cfg.ribbon_width)
    rng = np.random.default_rng(17)
    g = np.zeros((10, 5))
    g[:, 0] = w.start5[0] + rng.uniform(-45, 45, 10)
    g[:, 1] = w.start5[1] + rng.uniform(-45, 45, 10)
    g[:, 2] = rng.uniform(0, 2 * np.pi, 10)
    g[:, 3] = rng.uniform(0, 2, 10)
    g[:, 4] = w.start5[4]
    world = orc.World(w.cfg, w.grid, w.res, gauss=g)
    t0, dt, calls, init = 1000.0, 1e-3, 40, 256
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, t0, dt, calls, init, gauss=g)
        host = _run_cli(sc)
        rc, st, plan, itf, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init)
        assert rc == 0
        print({k: host[k] for k in host if k != "plan"})
        _compare(host, st, plan)


def test_speculative_expansion_changes_nothing_but_the_round_trips():
    """GpuAStarPlanner costs the children of the most promising open vertices ahead of their expansion (one device round
    trip for up to `speculation` vertices).  What the search pushes, pops and returns must not depend on it."""
    from path_planner_amd import workloads
    w = workloads.by_name("cfg3")
    outs = []
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        for spec in (1, 4, 32):
            sc = os.path.join(d, f"s{spec}.txt")
            _scenario(w, sc, mp, 1000.0, 1e-3, 60, 512, speculation=spec)
            outs.append(_run_cli(sc))
    base = outs[0]
    for o in outs[1:]:
        for k in base:
            if k in ("edges_costed", "wall_ms_median", "wall_ms_max"):
                continue
            assert o[k] == base[k], (k, o[k], base[k])
        assert o["edges_costed"] >= base["edges_costed"]
    assert base["expanded"] >= 10


def _read_search_dump(path):
    """The text format of the reference's search dump, tokenised the way its viewer does (visualizer.py:485-547): drop
    "():," and split on blanks; "Generated"/"Expanded" prefix; ribbon block between "Ribbons" and "End Ribbons"; otherwise
    State x y heading speed time f F g G h H tag [ids...]."""
    items, ribbons, in_ribbons = [], [], False
    for raw in open(path):
        line = raw
        for ch in "():,\n":
            line = line.replace(ch, "")
        tok = [t for t in line.split(" ")]
        while tok and tok[-1] == "":
            tok.pop()
        if not tok:
            continue
        kind = None
        if tok[0] in ("Expanded", "Generated"):
            kind, tok = tok[0], tok[1:]
        if tok[0] == "End" and tok[1] == "Ribbons":
            in_ribbons = False
        elif tok[0] == "Ribbons":
            in_ribbons = True
        elif in_ribbons:
            if tok[0] != "None":
                ribbons.append([float(tok[0]), float(tok[1]), float(tok[3]), float(tok[4])])
        elif tok[0] == "Trajectory":
            items.append({"tag": "trajectory-start"})
        elif tok[0] == "Incumbent":
            items.append({"tag": "incumbent", "f": float(tok[2])})
        else:
            assert tok[0] == "State", raw
            items.append({"kind": kind, "x": float(tok[1]), "y": float(tok[2]), "heading": float(tok[3]), "speed": float(tok[4]),
                          "time": float(tok[5]), "f": float(tok[7]), "g": float(tok[9]), "h": float(tok[11]), "tag": tok[12].lower(),
                          "ids": [int(t) for t in tok[13:]]})
    return items, ribbons


def test_search_dump_is_what_the_visualizer_reads():
    """PlannerConfig::setVisualizations + Visualizer (PlannerConfig.h:60-80, Visualizer.h): the dump of the search — start,
    incumbent, ribbons, samples, per-edge trajectories, generated / expanded vertices with their ancestry, goal plans — in the
    reference's text format, and a search that is not changed by being watched."""
    from path_planner_amd import workloads
    w = workloads.by_name("cfg3")
    t0, dt, calls, init = 1000.0, 1e-3, 24, 256
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, t0, dt, calls, init)
        plain = _run_cli(sc)
        dump = os.path.join(d, "search.txt")
        with open(sc, "a") as f:
            f.write(f"visualization_file {dump}\n")
        seen = _run_cli(sc)
        same = lambda r: {k: v for k, v in r.items() if not k.startswith("wall_ms")}
        assert same(plain) == same(seen)
        items, ribbons = _read_search_dump(dump)
    tags = [i["tag"] for i in items]
    assert tags.count("start") == seen["iterations"] and tags.count("incumbent") == seen["iterations"]
    assert len(ribbons) == len(w.ribbons4) * seen["iterations"]
    assert np.allclose(np.array(ribbons[:len(w.ribbons4)]), w.ribbons4, atol=1e-4)
    expanded = [i for i in items if i.get("kind") == "Expanded"]
    generated = [i for i in items if i.get("kind") == "Generated" and i["tag"] == "vertex"]
    assert len(expanded) == seen["expanded"]
    assert seen["generated"] <= len(generated) <= seen["generated"] + seen["iterations"]     # + the goal each aStar() returned
    # samples are listed every iteration; the last listing is the final sample set
    last_start = len(tags) - 1 - tags[::-1].index("start")
    assert tags[last_start:].count("sample") == seen["samples"]
    # ancestry: the last id is the vertex's own, the chain is as long as the vertex is deep, children of the root start with 1
    own = {}
    for g in generated:
        assert g["ids"][0] == 1 and abs(g["f"] - (g["g"] + g["h"])) <= 1e-3 * max(1.0, g["f"])
        own.setdefault(g["ids"][-1], g)
    assert len(own) >= seen["generated"] - seen["iterations"]
    deep = max(generated, key=lambda g: len(g["ids"]))
    assert len(deep["ids"]) >= 3 and all(i in own or i == 1 for i in deep["ids"])
    # trajectories: every edge streams one; its samples advance in time by a whole number of collision-check steps, cost
    # so far = start g + elapsed time + penalties, h = the START vertex's h
    inc_t = w.cfg.collision_checking_increment / w.cfg.max_speed
    per = int(1.0 / w.cfg.collision_checking_increment) + 1
    n_traj = tags.count("trajectory-start")
    assert n_traj >= seen["generated"] - 2 * seen["iterations"]
    checked = 0
    for a in range(len(items) - 2):
        if items[a]["tag"] == "trajectory" and items[a + 1]["tag"] == "trajectory":
            p, q = items[a], items[a + 1]
            assert abs((q["time"] - p["time"]) - per * inc_t) < 1e-5 and p["h"] == q["h"]
            dg = q["g"] - p["g"]
            pen = round((dg - per * inc_t) / 600.0)
            tol = 2e-5 * max(abs(p["g"]), abs(q["g"])) + 1e-4          # operator<< prints six significant digits
            assert pen >= 0 and abs(dg - per * inc_t - 600.0 * pen) < tol
            assert np.hypot(q["x"] - p["x"], q["y"] - p["y"]) <= per * inc_t * w.cfg.max_speed + 1e-4
            checked += 1
    assert checked > 100
    # a goal was found: its plan sampled once per second and the goal vertex itself
    assert seen["first_goal_iteration"] >= 0 and "goal" in tags and "plan" in tags


# ---------------------------------------------------------------------------------------------- late first goals
# Cluttered worlds found with tools/find_late_goal.py (CPU oracle): fine clutter and few initial samples make the first
# iterations exhaust the open list without reaching the horizon, so the FIRST goal comes several sample-doublings in.
# (name of the base configuration, blocked fraction, grid seed, blob size in cells, initial samples, first-goal iteration found)
LATE_GOALS = [("cfg2", 0.30, 106, 8, 16, 6), ("cfg2", 0.35, 111, 8, 32, 6), ("cfg3", 0.20, 112, 24, 8, 8)]


def _late_goal_workload(name, frac, gseed, blob):
    from path_planner_amd import workloads
    w = workloads.by_name(name)
    c = float(w.start5[0])
    w.grid = workloads.blob_grid(w.grid.shape[0], w.res, frac, gseed, (c, c), keep_free_radius=2.0, blob=blob)
    return w


@pytest.mark.parametrize("name,frac,gseed,blob,init,expect", LATE_GOALS)
def test_first_goal_iteration_matches_when_the_first_goal_comes_late(name, frac, gseed, blob, init, expect, prepass_route):
    """BASELINE's second metric on cases where it can differ: the first goal is found at iteration 6 to 8 (after as many
    doublings of the sample set), and the host planner must find it in the same iteration, with the same statistics and the
    same plan segment by segment."""
    import oracle as orc
    w = _late_goal_workload(name, frac, gseed, blob)
    orc.O.ppo_set_ribbon_width(w.cfg.ribbon_width)
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    t0, dt, calls = 1000.0, 1e-3, 400
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, t0, dt, calls, init)
        host = _run_cli(sc)
        rc, st, plan, itf, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init)
        assert rc == 0
        print(name, {k: host[k] for k in host if k != "plan"})
        assert st.first_goal_iteration == expect and st.first_goal_iteration >= 2      # the case is what the search tool found
        _compare(host, st, plan)


def test_brown_path_seeds_are_costed_and_pushed_like_the_reference():
    """PlannerConfig::useBrownPaths (AStarPlanner.cpp:40-43,99,150-162; RibbonManager::findNearStatesOnRibbons): the seeds on nearby
    ribbons are connected from the root at the coverage radius and both speeds at the start of every iteration.  The start is
    placed beside a ribbon so that seeds exist; same statistics and plan as the oracle's planner with the option on, and the
    option must change the search (otherwise the test would pass without the code)."""
    import oracle as orc
    from path_planner_amd import workloads
    w = workloads.by_name("cfg2")
    c = float(w.start5[0])
    w.start5 = np.array([c - 6.0, c + 4.0, 1.57, 2.5, 1.0])
    orc.O.ppo_set_ribbon_width(w.cfg.ribbon_width)
    seeds = np.zeros((16, 5))
    r4 = np.ascontiguousarray(w.ribbons4)
    assert orc.O.ppo_ribbons_near_states(r4.ctypes.data, len(r4), w.start5.ctypes.data, w.cfg.coverage_turning_radius, seeds.ctypes.data, 16) >= 1
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    t0, dt, calls, init = 1000.0, 1e-3, 40, 128
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        outs = {}
        for brown in (False, True):
            sc = os.path.join(d, f"s{int(brown)}.txt")
            _scenario(w, sc, mp, t0, dt, calls, init, brown=brown)
            host = _run_cli(sc)
            rc, st, plan, _, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init, use_brown_paths=brown)
            assert rc == 0
            print("brown", brown, {k: host[k] for k in host if k != "plan"})
            _compare(host, st, plan)
            outs[brown] = host
        assert outs[True]["generated"] != outs[False]["generated"] or outs[True]["edges_costed"] != outs[False]["edges_costed"]


def test_several_device_contexts_give_the_same_plan():
    """GpuAStarPlanner over a list of device contexts (one host thread per context; world and samples replicated, the open
    vertices of every batch dealt across them): same statistics and plan as on one context.  A 1-GPU box gives every context
    the same device — the dealing, the threads and the merge are what is exercised; 8 physical devices only change where the
    kernels run."""
    from path_planner_amd import workloads
    w = workloads.by_name("cfg3")
    outs = []
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        for devs in (None, [0, 0], [0, 0, 0, 0, 0]):
            sc = os.path.join(d, f"s{len(devs) if devs else 1}.txt")
            _scenario(w, sc, mp, 1000.0, 1e-3, 60, 512, speculation=16, devices=devs)
            outs.append(_run_cli(sc))
    base = outs[0]
    assert base["expanded"] >= 10
    for o in outs[1:]:
        for k in base:
            if k in ("edges_costed", "wall_ms_median", "wall_ms_max"):
                continue
            assert o[k] == base[k], (k, o[k], base[k])


def test_children_with_long_ribbon_lists_do_not_abort_the_plan():
    """TspPointRobotNoSplitAllRibbons on five parallel ribbons with the vehicle about to cross them all near their ends (open
    water, so nothing but the ribbons shapes the search): children carry up to ten pieces, more than the device's brute-force
    enumeration takes (8).  The reference enumerates any length; the host planner
    computes h of those children itself instead of dropping the plan (the oracle's exhaustive recursion cannot finish ten
    ribbons, so the check here is that a plan comes back and that the host path was actually taken; the value itself is
    checked against the oracle at sizes it can do in tests/test_host_cpu.py and test_gpu_parity.py)."""
    from path_planner_amd import workloads
    w = workloads.by_name("cfg3")
    w.cfg.heuristic = 1
    w.grid = np.zeros_like(w.grid)
    w.obst = None
    w.start5 = np.array([float(w.ribbons4[0][0]) + 2.0, float(w.ribbons4[0][1]) - 6.0, 0.0, 2.5, 1.0])
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, 1000.0, 1e-3, 300, 256)
        host = _run_cli(sc)
    print({k: host[k] for k in host if k != "plan"})
    assert "exception" not in host, host
    assert host["host_heuristics"] > 0 and host["plan_depth"] >= 1 and len(host["plan"]) >= 1


def test_ten_hertz_replan_loop_with_32_moving_obstacles(monkeypatch):
    """SURVEY 8(d) config 5 on one GPU: 120 consecutive plan() calls with a 100 ms real-time budget each, 8 192 initial samples doubling
    every iteration, the start advanced 0.1 s along the returned plan, the plan handed back as previousPlan, 32 moving obstacles on
    the config-3 grid.  Every cycle must return a plan BEFORE its deadline (Planner.h:42); the first (allocating) cycle is reported
    separately by plan_cli."""
    from path_planner_amd import workloads
    monkeypatch.delenv("PPGPU_PREPASS_MIN_EDGES", raising=False)      # the production setting
    w = workloads.config3()
    w.obst = workloads.obstacles(32, 3, 204.8, time=float(w.start5[4]))      # uniform in the map, as SURVEY 8(d) config 5 says: no free disc
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, float(w.start5[4]), 1e-3, 1, 8192, devices=[0, 0])      # two contexts (streams) on the one GPU: two round trips in flight
        with open(sc, "a") as f:
            f.write("time_remaining 0.1\nreplan 120 0.1\n")
        r = _run_cli(sc)
    print(r)
    assert r["replans"] == 120
    # A cycle can come back without a plan (the vehicle inside an obstacle's box, or heading into a cluster of them: every way forward
    # carries penalties, the heuristic stops guiding and 100 ms do not reach a goal at the 30 s horizon); the loop then does what
    # Executive::planLoop does — the third empty plan in a row halves the time horizon (executive.cpp:263-277) — so failures come in
    # runs of at most three per halving.  plan_cli also reports how many failed cycles started in collision.
    assert r["failed_plans"] <= 3 * (r["horizon_halvings"] + 1) and r["failed_plans"] <= 9, r
    assert r["mean_iterations"] >= 2 and r["mean_expanded"] >= 100 and r["cycles_with_a_goal"] >= 110
    # "Guaranteed to return before timeRemaining has elapsed" (Planner.h:42).  The deadline guard aims at the deadline minus a margin
    # (GpuContext::guardMargin) and does not start a round trip, a Brown-path re-cost or a sample doubling that cannot end before it;
    # the search tree's node array and every device buffer are sized before the loop (Stats::Budget counts what grows all the same).
    # 120 cycles: the 99th percentile is the third-worst cycle.
    assert r["wall_ms_p50"] < 100.0 and r["wall_ms_p99"] < 100.0 and r["wall_ms_max"] <= 103.0, r
    assert r["grid_uploads"] == 1, r            # the map did not change: the occupancy grid went to the device(s) in the first cycle only
    assert r["node_regrowths"] == 0 or r["worst_cycle"]["node_regrowths"] == 0, r


def test_a_failing_shard_does_not_leave_the_others_waiting():
    """ShardedIteration::run when one shard throws before the collective: every shard's host thread is joined, no shard enters
    ppgpu_allreduce_best (a rank that never joins an all-gather would leave the others waiting on their streams for ever), and the
    failure comes back as the call's exception — here with three contexts, shard 1 failing."""
    from path_planner_amd import workloads
    w = workloads.config2()
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, 1000.0, 1e-3, 10, 64, devices=[0, 0, 0])
        with open(sc, "a") as f:
            f.write("sharded_batch 3001 7\nfail_shard 1\n")
        out = subprocess.run([CLI, sc], capture_output=True, text=True, timeout=120)      # a hang would be the timeout
    assert out.returncode == 1, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert "injected failure in shard 1" in r["exception"], r


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_sample_sharded_iteration_in_the_cpp_product(devices):
    """ShardedIteration (SURVEY 8 e in the C++ host library): the iteration's batch split over the device contexts by sample index —
    ppgpu_sampler_skip, draw, cost, ppgpu_best_edge with the shard's index base — and combined with ONE collective,
    ppgpu_allreduce_best on the communicator ppgpu_comm_init_all made (one context: a 1-rank RCCL communicator, all a one-GPU box
    can form; three contexts on the one device: RCCL takes one rank per device, so the keys are combined on the host, and the
    split itself is what is checked).  Either way the incumbent is the unsharded launch's: same f bits, same sample, same
    configuration."""
    import torch                                     # before the HIP library: whichever libamdhip64 loads first serves both
    from path_planner_amd import api, sharding, workloads
    from path_planner_amd.types import RESULT_DTYPE, F_INFEASIBLE
    w = workloads.config2()
    attempts, seed = 3001, 7
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(w.grid, w.res, mp)
        sc = os.path.join(d, "s.txt")
        _scenario(w, sc, mp, 1000.0, 1e-3, 10, 64, devices=devices)
        with open(sc, "a") as f:
            f.write(f"sharded_batch {attempts} {seed}\n")
        r = _run_cli(sc)
    print(r)
    assert r["shards"] == len(devices) and r["agreed"]
    assert r["rccl_ranks"] == (1 if len(devices) == 1 else 0)
    # the unsharded launch through the C ABI
    ctx = api.Context(0)
    ctx.set_config(w.cfg); ctx.set_grid(w.grid, w.res); ctx.set_obstacles(w.obst)
    ctx.set_vertices(w.root(), w.ribbons4)
    ctx.sampler_init(w.bounds6, seed, w.ribbons4)
    n = ctx.sampler_add(attempts)
    assert sum(r["kept"]) == n and r["edges"] == 4 * n
    d_res = torch.zeros(4 * n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    ctx.cost_edges_dense(0, 1, 0, n, 0xF, d_res.data_ptr())
    ctx.synchronize()
    res = d_res.cpu().numpy().view(RESULT_DTYPE)
    key = sharding.local_best_key(res["f"], (res["flags"] & F_INFEASIBLE) == 0)
    assert int(key[0]) == r["best_f_bits"]
    # global edge id -> (shard, local edge) -> position in the unsharded list
    per = 4 * -(-attempts // len(devices))
    shard, local = r["best_edge"] // per, r["best_edge"] % per
    assert shard == r["best_shard"] and 4 * sum(r["kept"][:shard]) + local == int(key[1])
