"""-m gpu: a whole coverage mission through ppamd::Executive (the reference's planning harness, executive.cpp:43-305, ROS-free)
with a simulated vehicle that follows the published plans: the planner is called every 100 ms with the previous plan, the
vehicle's reported position covers the ribbons (Executive::updateCovered), and the loop ends when nothing is left to cover."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "path_planner_amd", "host", "mission_sim")


def test_two_ribbon_mission_completes():
    assert os.path.exists(SIM), "build the host tools: python -c 'import __graft_entry__ as g; g.build()'"
    from test_gpu_host_planner import _write_map
    grid = np.zeros((240, 240), dtype=np.uint8)
    grid[100:120, 150:170] = 1                      # an island off the survey lines
    with tempfile.TemporaryDirectory() as d:
        mp = os.path.join(d, "grid.map")
        _write_map(grid, 0.5, mp)
        sc = os.path.join(d, "m.txt")
        with open(sc, "w") as f:
            f.write("start 30 30 0.8 2.5 1000\n")
            f.write("ribbon 40 40 70 40\nribbon 70 46 40 46\n")
            f.write("obstacle 90 90 3.9 0.5 1000 5 10\n")
            f.write(f"map_file {mp}\n")
            # turningRadius coverageTurningRadius maxSpeed slowSpeed lineWidth k heuristic horizon tmin increment initialSamples brown gaussian ignore
            f.write("config 4 6 2.5 0.5 2 9 1 30 5 0.05 256 0 0 0\n")
            f.write("planning_time 0.1\nmax_seconds 150\n")
        out = subprocess.run([SIM, sc], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr[-2000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print(r)
        assert r["finished"] and not r["timed_out"], (r, out.stderr[-2000:])
        assert r["uncovered_length"] == 0.0
        assert r["plans_published"] >= 50 and r["empty_plans"] <= r["cycles"] // 10
        assert r["task_collision_penalty"] == 0.0


# ----------------------------------------------------------------------------------------------------------------------------
# The planning loop's decisions, cycle by cycle, against the oracle's restatement of Executive::planLoop.
TRACE = os.path.join(ROOT, "path_planner_amd", "host", "mission_trace")
ORACLE = os.path.join(ROOT, "oracle", "mission_oracle")

SCRIPT = """start 30 30 0.8 2.5 1000
ribbon 40 40 70 40
ribbon 70 46 40 46
obstacle 90 60 3.9 0.5 1000 5 10
map_file {map}
config 4 6 2.5 0.5 2 9 1 30 5 0.05 128 0 0 0
planning_time 0.1
clock 1000 0.004
max_cycles 40
at 6 clock_fault
at 10 teleport 60 80 1.0 2.5
at 20 teleport 30 60 2.0 2.5
at 26 horizon 30
"""


def write_scripted_mission(d):
    from test_gpu_host_planner import _write_map
    grid = np.zeros((240, 240), dtype=np.uint8)
    grid[100:120, 150:170] = 1
    mp = os.path.join(d, "grid.map")
    _write_map(grid, 0.5, mp)
    sc = os.path.join(d, "m.txt")
    with open(sc, "w") as f:
        f.write(SCRIPT.format(map=mp))
    return sc


def read_trace(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_planning_loop_decisions_match_the_oracle_executive():
    """A scripted mission (counting clock; a controller that keeps the vehicle on the plan; the vehicle displaced twice; one clock
    fault inside plan(); the horizon reconfigured) through ppamd::Executive + GpuAStarPlanner and through the oracle's restatement
    of executive.cpp:43-305 + the oracle's planner.  Per cycle: the state planned from (dead reckoning :114-118, the controller's
    answer :217-268), how much of the last plan is handed back (:144-146), the horizon after back-off (:270-287: three empty plans
    in a row halve it), the time budget (:189-190), the ribbons after covering up to the start (:186); then what plan() returned,
    and what the controller answered.  Integers identical, floating point within 1e-5."""
    assert os.path.exists(TRACE) and os.path.exists(ORACLE), "build: python -c 'import __graft_entry__ as g; g.build()'"
    with tempfile.TemporaryDirectory() as d:
        sc = write_scripted_mission(d)
        env = {k: v for k, v in os.environ.items() if k != "PPGPU_PREPASS_MIN_EDGES"}      # the production launch route
        got = subprocess.run([TRACE, sc], capture_output=True, text=True, timeout=600, env=env)
        assert got.returncode == 0, got.stdout[-2000:] + got.stderr[-2000:]
        want = subprocess.run([ORACLE, sc], capture_output=True, text=True, timeout=600)
        assert want.returncode == 0, want.stderr[-2000:]
    g, w = read_trace(got.stdout), read_trace(want.stdout)
    assert [r["k"] for r in g] == [r["k"] for r in w], ([r["k"] for r in g], [r["k"] for r in w])
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1.0)
    for a, b in zip(g, w):
        for key, vb in b.items():
            va = a[key]
            if isinstance(vb, list):
                assert len(va) == len(vb) and all(rel(x, y) <= 1e-5 for x, y in zip(va, vb)), (a, b)
            elif isinstance(vb, float) and not float(vb).is_integer():
                assert rel(va, vb) <= 1e-5, (key, a, b)
            else:
                assert va == vb, (key, a, b)
    # the script did exercise the rules it is there for
    cyc = [r for r in w if r["k"] == "cycle"]
    st = [r for r in w if r["k"] == "stats"]
    assert len(cyc) == 40 and w[-1]["empty_plans"] >= 6
    assert {r["time_horizon"] for r in cyc} == {30.0, 15.0}                              # halved after three failures, reconfigured, halved again
    assert any(r["previous_plan_legs"] >= 5 for r in cyc)                                # plans are handed back and grow
    assert st[6]["plan_legs"] == 0 and cyc[7]["previous_plan_legs"] == 0                 # the clock fault: exception -> empty plan
    assert cyc[11]["last_plan_achievable"] == 0 and cyc[11]["previous_plan_legs"] == 0   # displaced: the controller's answer is not on the plan
