"""The oracle's restatement of SamplingBasedPlanner::expand and AStarPlanner::plan: the reference tests'
structural assertions, plus determinism under the injected clock (PlannerConfig::setNowFunction)."""
import math

import numpy as np
import pytest

import oracle as orc
from path_planner_amd.types import H_MAX_DISTANCE, H_TSP_POINT_K, make_config
from path_planner_amd import workloads


def test_expand_test1_ribbons():
    """test_planner.cpp:1061-1082 ExpandTest1Ribbons: seed 9, box +-50, the generator's first state is the start
    (time 1), 1000 samples, one ribbon (0,10)-(0,30), empty Map: expand(root) leaves exactly 40 queue entries
    (4 nearest-endpoint edges + 2 radii x k=9 x 2 speeds), popped in non-decreasing f, then the queue is empty."""
    orc.O.ppo_set_ribbon_width(1.5)
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    b = np.array([-50, 50, -50, 50, 2.5, 2.5], dtype=np.float64)
    start, _ = orc.sampler_generate(b, 9, None, 0, 1)
    root = start[0].copy()
    root[4] = 1.0
    w = orc.World(cfg)
    rib = np.array([[0.0, 10.0, 0.0, 30.0]])
    f = np.zeros(64)
    gen, exp, kept = orc.C.c_uint64(), orc.C.c_uint64(), orc.C.c_uint64()
    n = orc.O.ppo_expand_once(w.h, 1, rib.ctypes.data, root.ctypes.data, b.ctypes.data, 9, 1, 1000, f.ctypes.data, 64,
                              orc.C.byref(gen), orc.C.byref(exp), orc.C.byref(kept))
    assert n == 40
    assert gen.value == 40 and exp.value == 1 and kept.value == 1000
    assert np.all(np.diff(f[:40]) >= 0)
    assert np.all(np.isfinite(f[:40])) and f[0] > 0


def test_expand_with_equal_speeds_and_radii_halves_the_fanout():
    """SamplingBasedPlanner.cpp:58-63: slowSpeed == maxSpeed and coverage radius == radius disable those variants."""
    orc.O.ppo_set_ribbon_width(1.5)
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE, slow_speed=2.5, coverage_turning_radius=8.0)
    b = np.array([-50, 50, -50, 50, 2.5, 2.5], dtype=np.float64)
    w = orc.World(cfg)
    rib = np.array([[0.0, 10.0, 0.0, 30.0]])
    root = np.array([3.0, -20.0, 0.3, 2.5, 1.0])
    f = np.zeros(64)
    n = orc.O.ppo_expand_once(w.h, 1, rib.ctypes.data, root.ctypes.data, b.ctypes.data, 5, 0, 500, f.ctypes.data, 64, None, None, None)
    assert n == 1 + 9      # one nearest-endpoint edge + k samples, one radius, one speed


def _plan(w, wl, budget_calls, **kw):
    # clock: now() = 1000 + calls * 1e-3; time_remaining = budget_calls * 1e-3 => a fixed number of now() polls
    return w.plan(wl.ribbons4, wl.start5, budget_calls * 1e-3, 1000.0, 1e-3, **kw)


def test_plan_is_deterministic_and_finds_a_goal_on_config1():
    """BASELINE config 1 (256x256 empty grid, 1 ribbon): same injected clock => identical stats, plan and
    first-goal iteration; the plan is continuous in time and starts at the start state."""
    wl = workloads.config1()
    w = orc.World(wl.cfg, wl.grid, wl.res, wl.obst)
    rc, st, plan, itf, _ = _plan(w, wl, 60, initial_samples=64)
    rc2, st2, plan2, itf2, _ = _plan(w, wl, 60, initial_samples=64)
    assert rc == rc2 == 0 and not st.threw
    for f in ("samples", "generated", "expanded", "iterations", "plan_f", "plan_depth", "first_goal_iteration", "plan_len"):
        assert getattr(st, f) == getattr(st2, f)
    assert np.array_equal(plan, plan2)
    assert st.first_goal_iteration >= 0 and st.plan_len >= 1
    assert st.iterations >= 1 and st.expanded >= 1
    assert plan[0, 9] == wl.start5[4]                      # first segment starts at the start state's time
    for a, b in zip(plan[:-1], plan[1:]):
        assert abs(a[10] - b[9]) < 1e-9                    # validatePlan (test_planner.cpp:27-41): contiguous in time
    # incumbent f never increases over iterations (AStarPlanner.cpp:109-117)
    best = itf[~np.isnan(itf)]
    assert np.all(np.diff(best) <= 1e-12)
    # seed = (unsigned long)(timeRemaining + now()) (AStarPlanner.cpp:15,33): a different clock origin changes the samples
    rc3, st3, plan3, _, _ = w.plan(wl.ribbons4, wl.start5, 60e-3, 2000.0, 1e-3, initial_samples=64)
    assert rc3 == 0 and (st3.generated != st.generated or not np.array_equal(plan3, plan))


def test_plan_with_done_ribbons_costs_nothing():
    """AStarPlanner.cpp:19 + Edge.cpp:93,198: with nothing left to cover every edge costs 0 and the root is the goal... after horizon."""
    cfg = make_config(start_state_time=5.0, heuristic=H_TSP_POINT_K, tsp_k=2)
    w = orc.World(cfg)
    rc, st, plan, itf, _ = w.plan(np.zeros((0, 4)), [0, 0, 0, 2.5, 5.0], 30e-3, 1000.0, 1e-3, initial_samples=32)
    assert rc == 0
    if st.plan_len:
        assert st.plan_f == 0.0


def test_checker_failures_come_back_as_a_return_code_not_an_abort():
    """No C++ exception may cross the ctypes boundary: a std::logic_error inside the oracle (here: a Dubins-TSP heuristic without
    a turning radius, RibbonManager.cpp:112-113) makes ppo_cost_edges return -2 with the message kept, instead of ending the
    test process in std::terminate."""
    import oracle as orc
    from path_planner_amd import workloads
    from path_planner_amd.types import edge_pack, H_TSP_DUBINS_ALL
    w = workloads.config1()
    w.cfg.heuristic = H_TSP_DUBINS_ALL
    w.cfg.heuristic_turning_radius = -1.0
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    cs = world.add_samples(w.bounds6, w.seed, w.ribbons4, 0, 8)
    e = edge_pack(np.zeros(4, dtype=np.uint64), np.arange(4), np.zeros(4, dtype=np.int64))
    with pytest.raises(AssertionError, match="unset turning radius"):
        world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e)
    assert orc.O.ppo_last_error() == b""          # reading clears it
    with pytest.raises(AssertionError, match="unset turning radius"):
        world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e, threads=2)


def test_skipping_the_heuristic_value_leaves_everything_else_alone():
    import oracle as orc
    from path_planner_amd import workloads
    from path_planner_amd.types import edge_pack
    w = workloads.config2(n_samples=64)
    world = orc.World(w.cfg, w.grid, w.res, w.obst)
    cs = world.add_samples(w.bounds6, w.seed, w.ribbons4, 0, 64)
    n = len(cs)
    e = edge_pack(np.zeros(4 * n, dtype=np.uint64), np.repeat(np.arange(n), 4), np.tile(np.arange(4), n))
    a, ca = world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e, stride=8)
    world.skip_heuristic_value(True)
    b, cb = world.cost_edges(w.root(), w.ribbons4, cs[:, 0], cs[:, 1], cs[:, 2], e, stride=8)
    world.skip_heuristic_value(False)
    assert np.array_equal(ca, cb)
    for f in a.dtype.names:
        if f not in ("h", "f"):
            assert np.array_equal(a[f], b[f], equal_nan=True) if a[f].dtype.kind == "f" else np.array_equal(a[f], b[f]), f
    feas = (a["flags"] & 1) == 0
    assert np.all(b["h"][feas] == 0) and np.any(a["h"][feas] > 0) and np.array_equal(b["f"][feas], b["g"][feas])


def test_oracle_executive_back_off_and_plan_reuse_rules():
    """oracle/mission_oracle (Executive::planLoop restated, executive.cpp:43-305) on the scripted mission of tests/test_gpu_mission.py,
    CPU only: three empty plans in a row halve the horizon and reset the count (:270-287); an exception inside plan() gives an empty
    plan and the loop goes on from dead reckoning (:191-195, :114-118); a controller answer off the plan drops the plan (:245-257);
    otherwise the remainder of the last plan is handed back (:144-146)."""
    import json, os, subprocess, tempfile
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_mission import ORACLE, write_scripted_mission, read_trace
    if not os.path.exists(ORACLE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(ORACLE), "mission_oracle"])
    with tempfile.TemporaryDirectory() as d:
        sc = write_scripted_mission(d)
        out = subprocess.run([ORACLE, sc], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
    tr = read_trace(out.stdout)
    cyc = [r for r in tr if r["k"] == "cycle"]
    st = [r for r in tr if r["k"] == "stats"]
    pub = {r["cycle"]: r for r in tr if r["k"] == "publish"}
    assert len(cyc) == len(st) == 40
    fails = 0
    horizon = 30.0
    for c, s in zip(cyc, st):
        assert c["empty_in_a_row"] == fails and c["time_horizon"] == horizon, (c, fails, horizon)
        if s["plan_legs"] == 0:
            assert c["cycle"] not in pub
            fails += 1
            if fails > 2:
                horizon = max(horizon / 2, 5.0)
                if horizon > 5.0:
                    fails = 0
        else:
            fails = 0
            assert c["cycle"] in pub
        if c["cycle"] == 26:
            horizon = 30.0                      # the scripted reconfiguration arrives after this cycle's plan()
    # plan reuse: the next cycle plans from the controller's answer and gets the last plan back, unless the answer was off the plan
    for c in cyc[1:]:
        prev = c["cycle"] - 1
        if prev in pub and prev not in (10, 20):
            assert c["from"] == pub[prev]["next"] and c["previous_plan_legs"] >= 1 and c["last_plan_achievable"] == 1
        elif prev in (10, 20):
            assert c["from"][:2] == pub[prev]["next"][:2] and c["previous_plan_legs"] == 0 and c["last_plan_achievable"] == 0
        else:
            assert c["previous_plan_legs"] == 0          # after an empty plan: dead reckoning, nothing to hand back
    assert st[6]["plan_legs"] == 0                       # the scripted clock fault
