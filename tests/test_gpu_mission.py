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
