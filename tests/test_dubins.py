"""Dubins solver: the oracle's restatement and the product's host C library (include/dubins.h,
path_planner_amd/csrc/dubins.c) against each other and against geometric properties.

The original third-party `dubins_curves` binary is absent and unpinned: general CSC/CCC
geometry is "parity unpinned" against it.  What pins the solver here: the reference tests' straight-line and
half-turn answers (tests/test_known_answers.py), and the properties below — resampling the end of the curve
reaches the target pose to 1e-9, the chosen word is the shortest of the six analytic words, angles stay in
[0, 2pi), prefixes sample consistently."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "path_planner_amd", "csrc")


class DubinsPath(C.Structure):
    _fields_ = [("qi", C.c_double * 3), ("param", C.c_double * 3), ("rho", C.c_double), ("type", C.c_int)]


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(CSRC, "libdubins.so")
    src = os.path.join(CSRC, "dubins.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-fPIC", "-ffp-contract=off", "-shared", src, "-o", so, "-lm"])
    L = C.CDLL(so)
    L.dubins_shortest_path.argtypes = [C.POINTER(DubinsPath), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
    L.dubins_path_length.argtypes = [C.POINTER(DubinsPath)]
    L.dubins_path_length.restype = C.c_double
    L.dubins_path_sample.argtypes = [C.POINTER(DubinsPath), C.c_double, C.POINTER(C.c_double)]
    L.dubins_extract_subpath.argtypes = [C.POINTER(DubinsPath), C.c_double, C.POINTER(DubinsPath)]
    return L


def _pairs(n, seed):
    rng = np.random.default_rng(seed)
    for i in range(n):
        q0 = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 2 * math.pi)])
        scale = [60.0, 10.0, 1.0][i % 3]      # far, medium and closer-than-a-turning-circle targets
        q1 = np.array([q0[0] + rng.uniform(-scale, scale), q0[1] + rng.uniform(-scale, scale), rng.uniform(0, 2 * math.pi)])
        yield q0, q1, [8.0, 16.0, 2.0][i % 3]


def _solve(lib, q0, q1, rho):
    p = DubinsPath()
    a = (C.c_double * 3)(*q0)
    b = (C.c_double * 3)(*q1)
    e = lib.dubins_shortest_path(C.byref(p), a, b, rho)
    return e, p


def test_host_library_equals_oracle_bitwise(lib):
    """Two independent implementations of the same published formulas, same libm: identical bits."""
    for q0, q1, rho in _pairs(3000, 1):
        e, p = _solve(lib, q0, q1, rho)
        eo, po = orc.dubins_shortest_path(q0, q1, rho)
        assert e == eo == 0
        assert p.type == int(po[7])
        assert list(p.param) == list(po[3:6])
        L = lib.dubins_path_length(C.byref(p))
        for frac in (0.0, 0.13, 0.5, 0.97, 1.0):
            q = (C.c_double * 3)()
            assert lib.dubins_path_sample(C.byref(p), L * frac, q) == 0
            eq, qo = orc.dubins_sample(po, L * frac)
            assert eq == 0 and list(q) == list(qo)


def test_end_of_curve_reaches_target_pose(lib):
    worst = 0.0
    for q0, q1, rho in _pairs(3000, 2):
        e, p = _solve(lib, q0, q1, rho)
        assert e == 0
        L = lib.dubins_path_length(C.byref(p))
        q = (C.c_double * 3)()
        assert lib.dubins_path_sample(C.byref(p), L, q) == 0 or lib.dubins_path_sample(C.byref(p), L - 1e-12, q) == 0
        dth = abs((q[2] - q1[2] + math.pi) % (2 * math.pi) - math.pi)
        err = max(abs(q[0] - q1[0]), abs(q[1] - q1[1]), dth)
        worst = max(worst, err)
        assert 0 <= q[2] < 2 * math.pi
    assert worst < 1e-9, worst


def test_shortest_of_the_six_words_and_first_wins_ties(lib):
    for q0, q1, rho in _pairs(2000, 3):
        e, p = _solve(lib, q0, q1, rho)
        best, best_w = math.inf, -1
        for w in range(6):
            out = np.zeros(3)
            if orc.O.ppo_dubins_word(w, np.ascontiguousarray(q0).ctypes.data, np.ascontiguousarray(q1).ctypes.data, rho, out.ctypes.data) == 0:
                c = out[0] + out[1] + out[2]
                assert out.min() >= 0 and out[0] < 2 * math.pi + 1e-12
                if c < best:
                    best, best_w = c, w
        assert best_w == p.type
        assert p.param[0] + p.param[1] + p.param[2] == best
        assert lib.dubins_path_length(C.byref(p)) >= math.hypot(q1[0] - q0[0], q1[1] - q0[1]) - 1e-9   # never shorter than the chord


def test_sample_is_continuous_and_heading_rate_bounded(lib):
    """The property the reference checks in AngleConsistencyTest(2) (tp:1122-1182): consecutive samples one
    collision-check step apart differ by at most step/rho (+1e-5) in heading and by about the step in position."""
    step = 0.05
    for q0, q1, rho in _pairs(60, 4):
        e, p = _solve(lib, q0, q1, rho)
        L = lib.dubins_path_length(C.byref(p))
        prev = None
        s = 0.0
        while s <= L:
            q = (C.c_double * 3)()
            assert lib.dubins_path_sample(C.byref(p), s, q) == 0
            if prev is not None:
                assert math.hypot(q[0] - prev[0], q[1] - prev[1]) <= step + 1e-9
                dth = abs((q[2] - prev[2] + math.pi) % (2 * math.pi) - math.pi)
                assert dth <= step / rho + 1e-9
            prev = (q[0], q[1], q[2])
            s += step


def test_out_of_range_sample_is_edubparam(lib):
    e, p = _solve(lib, [0, 0, 0], [10, 3, 1.0], 2.0)
    L = lib.dubins_path_length(C.byref(p))
    q = (C.c_double * 3)()
    assert lib.dubins_path_sample(C.byref(p), -1e-9, q) == 2
    assert lib.dubins_path_sample(C.byref(p), L * (1 + 1e-12) + 1e-9, q) == 2
    bad = DubinsPath()
    assert lib.dubins_shortest_path(C.byref(bad), (C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 0), 0.0) == 3   # EDUBBADRHO


def test_extract_subpath_is_a_prefix(lib):
    for q0, q1, rho in _pairs(200, 5):
        e, p = _solve(lib, q0, q1, rho)
        L = lib.dubins_path_length(C.byref(p))
        sub = DubinsPath()
        t = 0.37 * L
        assert lib.dubins_extract_subpath(C.byref(p), t, C.byref(sub)) == 0
        assert abs(lib.dubins_path_length(C.byref(sub)) - t) < 1e-9
        for frac in (0.0, 0.5, 0.999):
            a = (C.c_double * 3)()
            b = (C.c_double * 3)()
            assert lib.dubins_path_sample(C.byref(p), t * frac, a) == 0
            assert lib.dubins_path_sample(C.byref(sub), t * frac, b) == 0
            assert max(abs(a[i] - b[i]) for i in range(3)) < 1e-9
