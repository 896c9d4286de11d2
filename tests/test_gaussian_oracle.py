"""GaussianDynamicObstaclesManager (pp/src/common/dynamic_obstacles/GaussianDynamicObstaclesManager.{h,cpp}) as restated
in the oracle.  The reference file needs Eigen (absent here) and its only test prints values without asserting
(test_planner.cpp:230-238), so this model is PARITY UNPINNED against reference output: what pins the restatement is the
closed form of a bivariate normal density, evaluated independently here with numpy."""
import math

import numpy as np

import oracle as orc
from path_planner_amd.types import make_config


def _pdf(p, mean, cov):
    d = np.asarray(p, dtype=np.float64) - np.asarray(mean, dtype=np.float64)
    return float(math.exp(-0.5 * d @ np.linalg.inv(cov) @ d) / (2 * math.pi * math.sqrt(np.linalg.det(cov))))


def test_default_covariance_density_and_floor():
    cov = np.array([[30.0, 10.0], [10.0, 30.0]])
    w = orc.World(make_config(), gauss=[[0.0, 0.0, 0.0, 1.0, 1.0]])        # manager.update(1, 0, 0, 0, 1, 1), test_planner.cpp:232
    assert math.isclose(w.collision_exists(0.0, 0.0, 1.0), 1.0 / (2 * math.pi * math.sqrt(800.0)), rel_tol=1e-15)
    for i in range(10):                                                    # the grid GaussianDynamicObstacleTest1 prints
        for p in ((0.0, 10.0 * i), (10.0 * i, 10.0 * i), (10.0 * i, 0.0)):
            want = _pdf(p, (0.0, 0.0), cov)
            got = w.collision_exists(p[0], p[1], 1.0)
            if want < 1e-5:
                assert got == 0.0                                          # "questionable" floor, .cpp:11
            else:
                assert math.isclose(got, want, rel_tol=1e-13)
    assert w.collision_exists(0.0, 10.0, 1.0, strict=True) == w.collision_exists(0.0, 10.0, 1.0, strict=False)   # strict is ignored


def test_projection_and_sum_and_custom_covariance():
    # heading 0 = north: Yaw = pi/2, so after dt the mean moves by speed*dt along +y (Obstacle::project, .h:31-36)
    rows = [[0.0, 0.0, 0.0, 2.0, 1.0, 9.0, 1.0, 1.0, 4.0],
            [5.0, -3.0, math.pi / 2, 1.0, 0.0, 30.0, 10.0, 10.0, 30.0]]
    w = orc.World(make_config(), gauss=rows)
    t = 4.0
    m0 = (0.0 + 2.0 * 3.0 * math.cos(math.pi / 2), 0.0 + 2.0 * 3.0 * math.sin(math.pi / 2))
    m1 = (5.0 + 1.0 * 4.0 * math.cos(0.0), -3.0 + 1.0 * 4.0 * math.sin(0.0))
    for p in ((0.5, 5.0), (8.0, -2.0), (3.0, 1.0)):
        want = _pdf(p, m0, np.array([[9.0, 1.0], [1.0, 4.0]])) + _pdf(p, m1, np.array([[30.0, 10.0], [10.0, 30.0]]))
        assert math.isclose(w.collision_exists(p[0], p[1], t), want, rel_tol=1e-12)
