"""The product's host-side classes (path_planner_amd/host: RibbonManager over a flat ribbon array, table-driven TSP heuristics
with pruning, dense obstacle tables, DubinsWrapper) against the CPU oracle's restatement of the reference — on random inputs,
bit for bit, without a GPU.  The single-object primitives are pinned to the reference's own outputs in test_golden.py; this
file covers what only exists as a list operation or an algorithm (cover sequences, coverBetween, the five heuristics,
nearest endpoint, Brown-path seeds) and the places where the host code is organised differently from both the reference and
the oracle (index lists + pruning instead of list copies and exhaustive recursion)."""
import math

import numpy as np
import pytest

import hostlib as hl
import oracle as orc

H = hl.H


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def random_ribbons(rng, n, w, box=80.0, min_len=None):
    out = []
    while len(out) < n:
        r = rng.uniform(0, box, 4)
        if math.hypot(r[2] - r[0], r[3] - r[1]) >= (min_len if min_len is not None else 2 * w):
            out.append(r)
    return np.array(out).reshape(-1, 4)


@pytest.fixture(autouse=True)
def _width():
    yield
    H.pph_set_ribbon_width(1.5)
    orc.O.ppo_set_ribbon_width(1.5)


def set_width(w):
    H.pph_set_ribbon_width(w)
    orc.O.ppo_set_ribbon_width(w)


def test_cover_sequences_and_min_distance():
    rng = np.random.default_rng(11)
    for trial in range(60):
        w = [1.5, 2.0, 0.75][trial % 3]
        set_width(w)
        mine = theirs = random_ribbons(rng, 4, w)
        for step in range(50):
            if len(theirs):
                r = theirs[int(rng.integers(0, len(theirs)))]
                p = r[:2] + rng.uniform(0, 1) * (r[2:] - r[:2]) + rng.uniform(-1.2 * w, 1.2 * w, 2)
            else:
                p = rng.uniform(0, 80, 2)
            assert same(hl.ribbons_min_distance(mine, p[0], p[1]), orc.ribbons_min_distance(theirs, p[0], p[1]))
            strict = bool(step % 2)
            mine = hl.ribbons_cover(mine, p[0], p[1], strict)
            theirs = orc.ribbons_cover(theirs, p[0], p[1], strict)
            assert same(mine, theirs)


def test_cover_between_and_add():
    rng = np.random.default_rng(12)
    for trial in range(80):
        w = [1.5, 2.5][trial % 2]
        set_width(w)
        base = random_ribbons(rng, 3, w)
        a = base[0][:2] + rng.uniform(-2, 2, 2)
        b = base[0][2:] + rng.uniform(-2, 2, 2)
        if trial % 5 == 0:
            b = a + np.array([0.0, rng.uniform(1, 30)])          # vertical: atan(+-inf)
        for strict in (False, True):
            assert same(hl.ribbons_cover_between(base, a[0], a[1], b[0], b[1], strict),
                        orc.ribbons_cover_between(base, a[0], a[1], b[0], b[1], strict))
        # add() keeps out what is already shorter than the minimum length
        short = np.array([10.0, 10.0, 10.0 + 1.9 * w, 10.0])
        assert same(hl.ribbons_add(base, *short), orc.ribbons_add(base, *short))
        assert same(hl.ribbons_add(base, 1.0, 2.0, 40.0, 9.0), orc.ribbons_add(base, 1.0, 2.0, 40.0, 9.0))


@pytest.mark.parametrize("heuristic", [0, 1, 2, 3, 4])
def test_heuristics_equal_the_oracle_bit_for_bit(heuristic):
    """RibbonManager::approximateDistanceUntilDone, all five (RibbonManager.cpp:28-140, 234-248): the host's table + index-list
    search with pruning against the oracle's exhaustive recursion over copied lists."""
    rng = np.random.default_rng(100 + heuristic)
    set_width(1.5)
    checked = 0
    for trial in range(60):
        n = int(rng.integers(1, 6 if heuristic in (1, 3) else 7))
        ribs = random_ribbons(rng, n, 1.5, min_len=1.6)           # pieces down to w: length - 2w may be negative
        if trial % 7 == 0 and n >= 2:
            ribs[1, :2] = ribs[0, 2:]                              # pieces that share an endpoint: equal sort keys
        x, y, yaw = rng.uniform(0, 80), rng.uniform(0, 80), rng.uniform(0, 2 * math.pi)
        for K in ((2, 1, 3, 0) if heuristic in (2, 4) else (2,)):
            a = hl.ribbons_heuristic(ribs, heuristic, K, x, y, yaw, 8.0)
            b = orc.ribbons_heuristic(ribs, heuristic, K, x, y, yaw, 8.0)
            assert same(a, b), (heuristic, K, n, a, b)
            checked += 1
    assert checked >= 60
    assert hl.ribbons_heuristic(np.zeros((0, 4)), heuristic, 2, 1.0, 2.0) == 0.0   # done(): 0


def test_point_k_heuristic_beyond_the_device_limit():
    """Child lists longer than the device enumerates (12 ribbons for the K variant): the host computes the reference's value.
    14 pieces with K = 2 are 4^14 leaves for the reference's recursion; the oracle is capped at what finishes in seconds, the
    host's pruned search is checked against it there and must still answer quickly above."""
    import time
    rng = np.random.default_rng(5)
    set_width(1.5)
    for n in (9, 10, 11):
        ribs = random_ribbons(rng, n, 1.5, box=120.0)
        x, y = rng.uniform(0, 120, 2)
        assert same(hl.ribbons_heuristic(ribs, 2, 2, x, y), orc.ribbons_heuristic(ribs, 2, 2, x, y))
    ribs = random_ribbons(rng, 9, 1.5, box=120.0)
    assert same(hl.ribbons_heuristic(ribs[:7], 1, 0, 3.0, 4.0), orc.ribbons_heuristic(ribs[:7], 1, 0, 3.0, 4.0))
    t0 = time.time()
    for n in (14, 16):
        ribs = random_ribbons(rng, n, 1.5, box=150.0)
        v = hl.ribbons_heuristic(ribs, 2, 2, 10.0, 20.0)
        lower = sum(math.hypot(r[2] - r[0], r[3] - r[1]) - 3.0 for r in ribs)
        assert math.isfinite(v) and v >= lower
    assert time.time() - t0 < 20.0


def test_nearest_endpoint_projection_and_brown_seeds():
    rng = np.random.default_rng(21)
    set_width(1.5)
    n_seeds = 0
    for trial in range(200):
        ribs = random_ribbons(rng, int(rng.integers(1, 6)), 1.5, box=60.0)
        s = np.array([rng.uniform(0, 60), rng.uniform(0, 60), rng.uniform(0, 2 * math.pi), 2.5, 7.0])
        if trial % 4 == 0:       # close to an entry point: the "other end" branch
            r = ribs[0]
            d = (r[2:] - r[:2]) / np.linalg.norm(r[2:] - r[:2])
            s[:2] = r[:2] + d * 1.5 + rng.uniform(-0.5, 0.5, 2)
        rc_h, e_h = hl.ribbons_nearest_endpoint(ribs, s)
        rc_o, e_o = orc.ribbons_nearest_endpoint(ribs, s)
        assert rc_h == rc_o == 0 and same(e_h, e_o)
        p_h, p_o = s.copy(), s.copy()
        H.pph_ribbons_project(ribs.ctypes.data, len(ribs), p_h.ctypes.data)
        orc.O.ppo_ribbons_project(ribs.ctypes.data, len(ribs), p_o.ctypes.data)
        assert same(p_h, p_o)
        near_h = hl.ribbons_near_states(ribs, s, 16.0)
        out = np.zeros((64, 5))
        k = orc.O.ppo_ribbons_near_states(ribs.ctypes.data, len(ribs), s.ctypes.data, 16.0, out.ctypes.data, 64)
        assert same(near_h, out[:k])
        n_seeds += k
    assert n_seeds > 50
    assert hl.ribbons_nearest_endpoint(np.zeros((0, 4)), np.zeros(5))[0] == 1


def test_state_helpers_beyond_the_golden_file():
    """headingDifference / distanceTo against the reference's own object when it is at hand; interpolate and the two string
    forms against their definitions (State.cpp:27-41,95-121)."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 2 * math.pi), rng.uniform(0, 3), rng.uniform(0, 100)])
        b = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(0, 2 * math.pi), rng.uniform(0, 3), a[4] + rng.uniform(0.1, 10)])
        hd = H.pph_state_heading_difference(a.ctypes.data, b[2])
        assert same(hd, math.fmod(math.fmod(b[2] - a[2], 2 * math.pi) + 3 * math.pi, 2 * math.pi) - math.pi)
        if orc.REF is not None:
            assert same(hd, orc.REF.ref_state_heading_difference(a.ctypes.data, b[2]))
            assert same(H.pph_state_distance_to(a.ctypes.data, b[0], b[1]), orc.REF.ref_state_distance_to(a.ctypes.data, b[0], b[1]))
        t = a[4] + rng.uniform(0, 1) * (b[4] - a[4])
        out = np.zeros(5)
        H.pph_state_interpolate(a.ctypes.data, b.ctypes.data, t, out.ctypes.data)
        span, el = b[4] - a[4], t - a[4]
        want_h = a[2] + (hd / span) * el
        if want_h >= 2 * math.pi:
            want_h -= 2 * math.pi
        want = [a[0] + ((b[0] - a[0]) / span) * el, a[1] + ((b[1] - a[1]) / span) * el, want_h, a[3] + ((b[3] - a[3]) / span) * el, t]
        assert same(out, want)
    s = np.array([1.5, -2.25, math.pi / 2, 2.5, 10.0])
    buf = hl.C.create_string_buffer(512)
    H.pph_state_to_string(s.ctypes.data, 0, buf, 512)
    assert buf.value.decode() == "1.500000 -2.250000 90.000000 2.500000 10.000000"
    H.pph_state_to_string(s.ctypes.data, 1, buf, 512)
    assert buf.value.decode() == "1.500000 -2.250000 1.570796 2.500000 10.000000"


def test_obstacle_tables_update_forget_and_models():
    """Dense track tables: re-reporting a contact replaces it in place, forget() compacts, ignored contacts never enter; the
    binary count and the Gaussian density agree with the oracle's managers on the same rows."""
    from path_planner_amd.types import make_config
    rng = np.random.default_rng(9)
    rows = np.column_stack([rng.uniform(0, 100, 12), rng.uniform(0, 100, 12), rng.uniform(0, 2 * math.pi, 12), rng.uniform(0, 3, 12),
                            rng.uniform(0, 5, 12), rng.uniform(2, 10, 12), rng.uniform(5, 30, 12)])
    hm = hl.Obstacles()
    for i, r in enumerate(rows):
        hm.update(7 + i, *r)
    hm.update(7 + 3, *rows[3])                          # again: same slot
    hm.forget(7 + 5)
    keep = [i for i in range(12) if i != 5]
    got = hm.device_rows(7)
    assert sorted(map(tuple, got)) == sorted(map(tuple, rows[keep]))
    w = orc.World(make_config(), obst=got)
    hits = 0
    for _ in range(3000):
        i = int(rng.integers(0, len(got)))
        t = rng.uniform(0, 30)
        yaw = math.pi / 2 - got[i, 2]
        x = got[i, 0] + got[i, 3] * (t - got[i, 4]) * math.cos(yaw) + rng.uniform(-12, 12)
        y = got[i, 1] + got[i, 3] * (t - got[i, 4]) * math.sin(yaw) + rng.uniform(-12, 12)
        for strict in (False, True):
            c = hm.collision_exists(x, y, t, strict)
            assert c == w.collision_exists(x, y, t, strict)
            hits += c > 0
    assert hits > 200
    # Gaussian model: default and explicit covariance, against the oracle in the same row order
    g = hl.Obstacles(gaussian=True)
    grows = []
    for i in range(6):
        r = [rng.uniform(0, 60), rng.uniform(0, 60), rng.uniform(0, 2 * math.pi), rng.uniform(0, 2), 1.0]
        if i % 2:
            cov = [20.0 + i, 3.0, 3.0, 12.0 + i]
            g.update_gaussian(50 + i, *r, cov)
        else:
            cov = [30.0, 10.0, 10.0, 30.0]
            g.update(50 + i, *r)
        grows.append(r + cov)
    assert np.array_equal(g.device_rows(9), np.array(grows))
    wg = orc.World(make_config(), gauss=np.array(grows))
    nonzero = 0
    for _ in range(2000):
        x, y, t = rng.uniform(0, 60), rng.uniform(0, 60), rng.uniform(0, 20)
        a, b = g.collision_exists(x, y, t), wg.collision_exists(x, y, t)
        assert same(a, b), (a, b)
        nonzero += a > 0
    assert nonzero > 100


def test_dubins_wrapper_solve_and_sample():
    """DubinsWrapper::set / sample incl. the retry 1e-5 short of the end and the refusal outside the window, against the oracle's
    wrapper on the same states (both sit on the same C solver semantics: csrc/dubins.c == the oracle's, tests/test_dubins.py)."""
    rng = np.random.default_rng(31)
    for _ in range(400):
        a = np.array([rng.uniform(0, 50), rng.uniform(0, 50), rng.uniform(0, 2 * math.pi), 2.5, 3.0])
        b = np.array([rng.uniform(0, 50), rng.uniform(0, 50), rng.uniform(0, 2 * math.pi), 2.5, 0.0])
        rho = [8.0, 16.0][int(rng.integers(0, 2))]
        path = np.zeros(8)
        end_h = H.pph_wrapper_solve(a.ctypes.data, b.ctypes.data, rho, path.ctypes.data)
        out_o, end_o = np.zeros(5), orc.C.c_double()
        for frac in (0.0, rng.uniform(0, 1), 1.0, 1.2):
            t = a[4] + frac * (end_h - a[4])
            rc_o = orc.O.ppo_wrapper_sample(a.ctypes.data, b.ctypes.data, rho, -1.0, t, out_o.ctypes.data, orc.C.byref(end_o))
            assert same(end_h, end_o.value)
            out_h = np.zeros(5)
            rc_h = H.pph_wrapper_sample(path.ctypes.data, 2.5, a[4], end_h, t, out_h.ctypes.data)
            assert rc_h == rc_o == (1 if frac > 1.0 else 0)
            if rc_h == 0:
                assert same(out_h, out_o), (out_h, out_o)


def test_dump_and_uncovered_length_formats():
    set_width(1.5)
    ribs = np.array([[0.0, 0.0, 10.0, 0.0], [5.5, 1.25, 5.5, 30.75]])
    buf = hl.C.create_string_buffer(1024)
    H.pph_ribbons_dump(ribs.ctypes.data, 2, buf, 1024)
    assert buf.value.decode() == "Ribbons: \n(0, 0) -> (10, 0) with length 10\n(5.5, 1.25) -> (5.5, 30.75) with length 29.5\n"
    H.pph_ribbons_dump(None, 0, buf, 1024)
    assert buf.value.decode() == "Ribbons: \nNone\n"
    # the reference sums ribbon lengths into an int, truncating after every ribbon (RibbonManager.cpp:413-417)
    assert H.pph_ribbons_total_uncovered_length(ribs.ctypes.data, 2) == 39.0
