"""The reference's own unit-test assertions for this path, transcribed as known answers for the CPU
oracle (SURVEY.md section 8 c).  Source: /root/reference/path_planner/test/planner/test_planner.cpp
(`tp:` below).  None of those gtests is runnable here (gtest, ROS message headers and the
third-party dubins_curves package are absent), so their inputs and expected values are restated.

Staleness found while transcribing (documented, not papered over):
  * tp:455-486, 498-510, 547-562 (RibbonsTest1/2/3/4, RibbonTest7) expect heuristic values WITHOUT the
    `- 2 * Ribbon::RibbonWidth` per-ribbon term that RibbonManager.cpp:60-63, 87-90 and 241 now subtract
    (the test file itself says "TODO! -- subtract 2* min ribbon length", tp:458).  tp:924-953
    (VertexTests2/3) DO include `- 2 * Ribbon::minLength()` and pin the current code.  The stale cases are
    asserted here as `value_in_test - 2 * w * n_ribbons` wherever the summed branch is the active one.
"""
import math

import numpy as np
import pytest

import oracle as orc
from path_planner_amd.types import (F_INFEASIBLE, H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K, edge_pack, make_config)
from path_planner_amd.workloads import root_vertex

W = 1.5  # Ribbon::RibbonWidth default (Ribbon.cpp:4)


@pytest.fixture(autouse=True)
def _ribbon_width():
    orc.O.ppo_set_ribbon_width(W)
    yield
    orc.O.ppo_set_ribbon_width(W)


def double_eq(a, b):
    """gtest EXPECT_DOUBLE_EQ: within 4 ULPs."""
    return abs(a - b) <= 4 * np.spacing(max(abs(a), abs(b), 1e-300))


def cost_edge(cfg, root5, ribbons4, target3, cfgbits=0, heuristic=None, root_speed=None):
    """Vertex::makeRoot + Vertex::connect(root, State) + Edge::computeTrueCost through the oracle."""
    world = orc.World(cfg)
    rib = np.asarray(ribbons4, dtype=np.float64).reshape(-1, 4)
    v = root_vertex(root5[0], root5[1], root5[2], root5[3], root5[4], rib)
    e = edge_pack(np.array([0], dtype=np.uint64), np.array([0], dtype=np.uint64), np.array([cfgbits], dtype=np.uint64))
    res, child = world.cost_edges(v, rib, [target3[0]], [target3[1]], [target3[2]], e, stride=8)
    return res[0], child[0]


# ------------------------------------------------------------------ dynamic obstacles
def test_binary_dynamic_obstacles_truth_table():
    """tp:202-216 BinaryDynamicObstaclesTest1: update(1, 42, 42, 0, 1, 1, 5, 15), non-strict."""
    world = orc.World(make_config(), obst=[[42, 42, 0, 1, 1, 5, 15]])
    ce = lambda x, y, t: world.collision_exists(x, y, t, strict=False)
    assert ce(42, 42, 1) == 1
    assert ce(42, 49, 1) == 1
    assert ce(42, 50, 1) == 0
    assert ce(44, 42, 1) == 1
    assert ce(45, 42, 1) == 0
    # projected 10 s ahead at 1 m/s due north
    assert ce(42, 52, 11) == 1
    assert ce(42, 59, 11) == 1
    assert ce(42, 60, 11) == 0
    assert ce(44, 52, 11) == 1
    assert ce(45, 52, 11) == 0


def test_base_obstacle_manager_returns_zero():
    """DynamicObstaclesManager.h:23: the base class never collides."""
    world = orc.World(make_config())
    assert world.collision_exists(42, 42, 1, True) == 0


# ------------------------------------------------------------------ Dubins wrapper
def test_simple_dubins_half_turn():
    """tp:441-449 SimpleDubinsTest: (0,0,h=0) -> (2*rho,0,h=pi), rho 8, speed 2 => end time 1 + pi*rho/speed."""
    radius, speed = 8.0, 2.0
    s1 = np.array([0, 0, 0, speed, 1.0])
    s2 = np.array([2 * radius, 0, math.pi, speed, 0.0])
    out = np.zeros(5)
    end = orc.C.c_double()
    rc = orc.O.ppo_wrapper_sample(s1.ctypes.data, s2.ctypes.data, radius, -1.0, 1.0, out.ctypes.data, orc.C.addressof(end))
    assert rc == 0
    assert abs(end.value - (radius * math.pi / speed + 1)) < 1e-5


def test_make_plan_straight_line():
    """tp:858-877 MakePlanTest: 5 m straight at 1 m/s, rho 2 => approx cost 5, samples start/end on the states."""
    s1 = np.array([0, 0, 0, 1.0, 1.0])
    s2 = np.array([0, 5, 0, 1.0, 6.0])
    e, p = orc.dubins_shortest_path([0, 0, orc.yaw(0)], [0, 5, orc.yaw(0)], 2.0)
    assert e == 0
    length = (p[3] + p[4] + p[5]) * p[6]
    assert double_eq(length / 1.0, 5)
    out = np.zeros(5)
    end = orc.C.c_double()
    assert orc.O.ppo_wrapper_sample(s1.ctypes.data, s2.ctypes.data, 2.0, -1.0, 1.0, out.ctypes.data, orc.C.addressof(end)) == 0
    assert out[0] == 0 and out[1] == 0 and out[2] == 0          # isCoLocated with s1
    assert end.value - 1.0 >= 5 - 1e-12
    assert orc.O.ppo_wrapper_sample(s1.ctypes.data, s2.ctypes.data, 2.0, -1.0, 6.0, out.ctypes.data, orc.C.addressof(end)) == 0
    assert math.hypot(out[0] - 0, out[1] - 5) < 1e-5


# ------------------------------------------------------------------ edges
def test_compute_edge_cost():
    """tp:879-893 ComputeEdgeCostTest: 5 m straight at 2.5 m/s from t=1 => child time 3, true == approx."""
    cfg = make_config(start_state_time=1.0)
    # The gtest builds the root with an EMPTY RibbonManager; with the current Edge.cpp:93,198
    # (`ribbonManagerStartedDone` => t = 0) such an edge costs 0, so `EXPECT_DOUBLE_EQ(c, a)` (a = 2) is stale.
    # Both behaviours are pinned: time 3 either way; cost 0 with no ribbons, cost == approx with work left.
    r, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], [], [0, 5, 0])
    assert double_eq(r["end_time"], 3)
    assert double_eq(r["approx_cost"], 2)
    assert r["true_cost"] == 0.0
    assert not (r["flags"] & F_INFEASIBLE)
    r, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], orc.ribbons_add([], 50, 50, 60, 50), [0, 5, 0])
    assert double_eq(r["end_time"], 3)
    assert double_eq(r["true_cost"], r["approx_cost"])


def test_vertex_tests1():
    """tp:907-923 VertexTests1: (5,5,h=pi) -> (5,-20,h=pi): approx 10 s, true == approx == g == t - 1, f = g + h."""
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    rib = orc.ribbons_add([], 50, 50, 60, 50)
    r, child = cost_edge(cfg, [5, 5, math.pi, 2.5, 1], rib, [5, -20, math.pi])
    assert double_eq(r["approx_cost"], 10)
    assert double_eq(r["true_cost"], r["approx_cost"])
    assert double_eq(r["true_cost"], r["g"])
    assert double_eq(r["g"], r["end_time"] - 1)
    h_expected = orc.ribbons_heuristic(rib, H_MAX_DISTANCE, 0, r["end_x"], r["end_y"]) / 2.5
    assert double_eq(r["h"], h_expected)
    assert double_eq(r["f"], r["true_cost"] + r["h"])


def test_vertex_tests2_tsp_heuristic():
    """tp:924-939 VertexTests2: h = (|(5,-20)-(30,30)| + 20*sqrt(2) + 10 + 50 - 2*minLength) / 2.5."""
    cfg = make_config(start_state_time=1.0, heuristic=H_TSP_POINT_ALL)
    rib = orc.ribbons_add(orc.ribbons_add([], 30, 30, 50, 50), 50, 60, 100, 60)
    r, _ = cost_edge(cfg, [5, 5, math.pi, 2.5, 1], rib, [5, -20, math.pi])
    assert double_eq(r["approx_cost"], r["true_cost"])
    d = math.hypot(r["end_x"] - 30, r["end_y"] - 30)
    assert double_eq(r["h"], (d + 20 * math.sqrt(2) + 10 + 50 - 2 * (2 * W)) / 2.5)


def test_vertex_tests3_max_distance_heuristic():
    """tp:940-953 VertexTests3: h = (|(5,-20)-(30,30)| + 20*sqrt(2) + 50 - 2*minLength) / 2.5."""
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    rib = orc.ribbons_add(orc.ribbons_add([], 30, 30, 50, 50), 50, 60, 100, 60)
    r, _ = cost_edge(cfg, [5, 5, math.pi, 2.5, 1], rib, [5, -20, math.pi])
    d = math.hypot(r["end_x"] - 30, r["end_y"] - 30)
    assert double_eq(r["h"], (d + 20 * math.sqrt(2) + 50 - 2 * (2 * W)) / 2.5)


def test_edge_truncation():
    """tp:1184-1207 EdgeTruncation: 10 m => 4 s both ways; 100 m => approx 40, true == horizon 30, child moved."""
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    rib = orc.ribbons_add([], 100, 0, 100, 10)
    r1, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], rib, [0, 10, 0])
    assert double_eq(r1["approx_cost"], 4) and double_eq(r1["true_cost"], 4)
    assert math.hypot(r1["end_x"] - 0, r1["end_y"] - 10) < 1e-10
    r2, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], rib, [0, 100, 0])
    assert double_eq(r2["approx_cost"], 40)
    # horizon + the 1e-12 fudge of Edge.cpp:90 (the gtest's EXPECT_DOUBLE_EQ(d, 30) is 4-ULP tight and cannot
    # hold with that fudge; DifferentSpeedsCoverageTest uses 1e-5 for the same quantity)
    assert abs(r2["true_cost"] - 30) <= 2e-12 and r2["true_cost"] > 30
    assert not (r2["end_x"] == 0 and r2["end_y"] == 10)


def test_different_speeds_coverage():
    """tp:1102-1120 DifferentSpeedsCoverageTest: 30 m along a ribbon; fast g = 12, slow truncated at the horizon."""
    cfg = make_config(start_state_time=1.0, heuristic=H_MAX_DISTANCE)
    rib = orc.ribbons_add([], 0, 0, 0, 30)
    fast, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], rib, [0, 30, 0], cfgbits=0)
    slow, _ = cost_edge(cfg, [0, 0, 0, 2.5, 1], rib, [0, 30, 0], cfgbits=2)
    assert double_eq(fast["g"], 30 / 2.5)
    assert abs(slow["g"] - 30.0) < 1e-5
    assert fast["g"] < slow["g"]
    assert fast["f"] < slow["f"]


# ------------------------------------------------------------------ ribbons
def test_ribbon_split():
    """tp:488-496 RibbonSplitTest."""
    r = np.array([40.0, 100.0, -70.0, -120.0])
    f = np.zeros(4)
    orc.O.ppo_ribbon_split(r.ctypes.data, 0.0, 0.0, 0, f.ctypes.data)
    assert math.hypot(f[2] - f[0], f[3] - f[1]) < 3          # (0,0) is > 1.5 m from the line: empty ribbon
    f2 = np.zeros(4)
    orc.O.ppo_ribbon_split(r.ctypes.data, -10.0, 0.0, 0, f2.ctypes.data)
    assert (f2[2], f2[3]) == (-10.0, 0.0)
    assert (f2[0], f2[1]) == (40.0, 100.0)
    assert (f2[2], f2[3]) == (r[0], r[1])


@pytest.mark.parametrize("heuristic,K", [(H_MAX_DISTANCE, 0), (H_TSP_POINT_ALL, 0), (H_TSP_POINT_K, 2)])
def test_ribbons_heuristic_values(heuristic, K):
    """tp:455-470 RibbonsTest1, :472-486 RibbonsTest2, :547-562 RibbonTest7 (stale by 2*w per ribbon, see module doc)."""
    one = orc.ribbons_add([], 0, 0, 1000, 0)
    two = orc.ribbons_add(one, 0, 20, 1000, 20)
    h = lambda rib, x, y: orc.ribbons_heuristic(rib, heuristic, K, x, y)
    s2 = math.sqrt(2) * 100
    if heuristic == H_MAX_DISTANCE:
        # max(sum(len - 2w) + nearest endpoint, farthest endpoint)
        assert double_eq(h(one, 0, 0), 1000)            # farthest endpoint wins: value in the test stands
        assert double_eq(h(one, -100, 0), 1100)
        assert double_eq(h(one, 0, 1000), 2000 - 2 * W)
        assert double_eq(h(one, 1000, 1000), 2000 - 2 * W)
        assert double_eq(h(one, 100, 100), 1000 + s2 - 2 * W)
        assert double_eq(h(two, 0, 0), 2000 - 4 * W)
        assert double_eq(h(two, -100, 0), 2100 - 4 * W)
        assert double_eq(h(two, 0, 1000), 2980 - 4 * W)
        assert double_eq(h(two, 1000, 1000), 2980 - 4 * W)
        assert double_eq(h(two, 100, 120), 2000 + s2 - 4 * W)
    else:
        assert double_eq(h(one, 0, 0), 1000 - 2 * W)
        assert double_eq(h(one, -100, 0), 1100 - 2 * W)
        assert double_eq(h(one, 0, 1000), 2000 - 2 * W)
        assert double_eq(h(one, 1000, 1000), 2000 - 2 * W)
        assert double_eq(h(one, 100, 100), 1000 + s2 - 2 * W)
        assert double_eq(h(two, 0, 0), 2020 - 4 * W)
        assert double_eq(h(two, -100, 0), 2120 - 4 * W)
        assert double_eq(h(two, 0, 1000), 3000 - 4 * W)
        assert double_eq(h(two, 1000, 1000), 3000 - 4 * W)
        assert double_eq(h(two, 100, 120), 2020 + s2 - 4 * W)


def test_ribbons_cover_then_heuristic():
    """tp:498-510 RibbonsTest3/4 (stale by 2*w): cover(2,0) leaves [2,1000]; cover(1,1) leaves [1,1000]."""
    one = orc.ribbons_add([], 0, 0, 1000, 0)
    r3 = orc.ribbons_cover(one, 2, 0, False)
    assert np.array_equal(r3, [[2, 0, 1000, 0]])
    assert double_eq(orc.ribbons_heuristic(r3, H_TSP_POINT_ALL, 0, 2, 0), 998 - 2 * W)
    r4 = orc.ribbons_cover(one, 1, 1, False)
    assert np.array_equal(r4, [[1, 0, 1000, 0]])
    assert double_eq(orc.ribbons_heuristic(r4, H_TSP_POINT_ALL, 0, 1, 0), 999 - 2 * W)


def test_get_nearest_endpoint():
    """tp:703-713 RibbonManagerGetNearestEndpointTest."""
    one = orc.ribbons_add([], 10, 10, 20, 10)
    rc, s = orc.ribbons_nearest_endpoint(one, [0, 0, 0, 0, 0])
    # the gtest asserts x == 10 / 20 / 2.66... with EXPECT_DOUBLE_EQ, which the `s.move(minLength/2 + 1e-5)` and
    # `ret.move(-minLength/2 + 1e-5)` of RibbonManager.cpp:166,173 cannot satisfy: stale expectations (they predate
    # the pull-in).  What is pinned here is WHICH endpoint is chosen and the pull-in distance.
    assert rc == 0
    assert abs(s[0] - 11.50001) < 1e-9 and abs(s[1] - 10) < 1e-9
    rc, s = orc.ribbons_nearest_endpoint(one, [10, 10, 0, 0, 0])
    # within minLength of the start: "we actually want the state at the other end", pulled back 1.5 - 1e-5
    assert abs(s[0] - (20 - 1.5 + 1e-5)) < 1e-9
    two = orc.ribbons_add(one, 2.6625366957003918, 60, 7.8363094365852275, 60)
    rc, s = orc.ribbons_nearest_endpoint(two, [7.8363094365852275, 60, 4.7123889803846897, 2.5, 83.397109423209002])
    assert abs(s[0] - (2.6625366957003918 + 1.5 - 1e-5)) < 1e-9


def test_heuristic_consistency_along_ribbon():
    """tp:564-588 HeuristicConsistency1: driving a 75 m ribbon dead ahead at 2.5 m/s, t + h stays 31."""
    for heuristic in (H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K):
        rib = orc.ribbons_add([], 0, 0, 0, 75)
        s1 = np.array([0, 0, 0, 2.5, 1.0])
        s2 = np.array([0, 75, 0, 2.5, 31.0])
        t = 1.0
        out = np.zeros(5)
        end = orc.C.c_double()
        checked = 0
        while t <= 31.0:
            assert orc.O.ppo_wrapper_sample(s1.ctypes.data, s2.ctypes.data, 8.0, -1.0, t, out.ctypes.data, orc.C.addressof(end)) == 0
            rib = orc.ribbons_cover(rib, out[0], out[1], False)
            if len(rib) == 0:
                break
            h = orc.ribbons_heuristic(rib, heuristic, 2, out[0], out[1]) / 2.5
            # MaxDistance: the farthest-endpoint branch wins and t + h == 31 exactly as the gtest says; the TSP
            # variants carry the -2w term the gtest predates: t + h == 31 - 2w/2.5
            expect = 31.0 if heuristic == H_MAX_DISTANCE else 31.0 - 2 * W / 2.5
            assert abs(t + h - expect) < 1e-9
            checked += 1
            t += 1
        assert checked >= 25


# ------------------------------------------------------------------ edges whose curve is given
def test_wrapper_edge_equals_state_edge_and_truncates():
    """Vertex::connect(start, DubinsWrapper, coverageAllowed) (Vertex.cpp:28-36, Edge.cpp:208-216) with the shortest path to a
    state is the same edge as connect(start, state); with updateEndTime (DubinsWrapper.cpp:100-104) it stops early; a curve that
    starts after the vertex's first step is infeasible with no step counted (the sample throws inside Edge.cpp:126-133)."""
    from path_planner_amd.types import WRAPPER_EDGE_DTYPE
    cfg = make_config(start_state_time=1.0)
    rib = orc.ribbons_add([], 0, 10, 0, 40)
    root5 = [0, 0, 0, 2.5, 1]
    tgt = [3.0, 30.0, 0.4]
    ref, refchild = cost_edge(cfg, root5, rib, tgt)
    world = orc.World(cfg)
    v = root_vertex(*root5, np.asarray(rib).reshape(-1, 4))
    q0 = [0.0, 0.0, orc.O.ppo_state_yaw(0.0)]
    q1 = [tgt[0], tgt[1], orc.O.ppo_state_yaw(tgt[2])]
    err, p8 = orc.dubins_shortest_path(q0, q1, cfg.turning_radius)
    assert err == 0
    end = orc.O.ppo_wrapper_fill_end_time(p8.ctypes.data, 2.5, 1.0)

    def wedge(start, stop):
        return np.array([(0, 0, p8[0:3], p8[3:6], p8[6], int(p8[7]), 0, 2.5, start, stop)], dtype=WRAPPER_EDGE_DTYPE)

    r, child = world.cost_wrapper_edges(v, rib, wedge(1.0, end), stride=8)
    assert r[0].tobytes() == ref.tobytes() and child[0].tobytes() == refchild.tobytes()
    cut = 1.0 + 0.5 * (end - 1.0)
    r, _ = world.cost_wrapper_edges(v, rib, wedge(1.0, cut), stride=8)
    assert r["end_time"][0] == cut and not (r["flags"][0] & F_INFEASIBLE)
    assert double_eq(r["approx_cost"][0], cut - 1.0) and (r["info"][0] >> 16) < (ref["info"] >> 16)
    r, _ = world.cost_wrapper_edges(v, rib, wedge(1.5, orc.O.ppo_wrapper_fill_end_time(p8.ctypes.data, 2.5, 1.5)), stride=8)
    assert (r["flags"][0] & F_INFEASIBLE) and (r["info"][0] >> 16) == 0
