#!/usr/bin/env python3
"""GPU box: randomized differential test of whole plan() calls — the C++ host planner (plan_cli, injected clock) against the
oracle's restatement of AStarPlanner::plan on random worlds: grid, obstacles, ribbons, heuristic, speeds, radii, budget, and
a second cycle that hands the first plan back (previous-plan re-costing).

Verdict per plan() call:
  ok        identical samples / iterations / expansions / generated / first-goal iteration / depth, costs within 1e-5, the same
            plan segment by segment;
  tie       all of the above except that ANOTHER plan of the same cost came back, or that `generated` / `expanded` moved by a few
            (a vertex pruned against the incumbent on one side only).  Every such case must be explained by the two
            edge dumps (classify_tie): both searches consume the same number of edges in the same order, no infeasible flag
            differs, and somewhere upstream the two sides disagree in the last digits of a curve — the reference's own
            `distance - 1e-5` retry (DubinsWrapper.cpp:39-42) fired on one side only, or two Dubins words of exactly equal length
            were told apart by the last bit of libm, or the Dubins problem itself is degenerate (collinear poses, a word on the
            edge of existing: the glibc solver returns the host's curve when its input is moved by 1e-11) — after
            which vertices of (near-)equal f pop in another order;
  MISMATCH  anything else (a failed check is named).
usage: tools/fuzz_plan.py [rounds] [seed] [round ...]     (tests/test_gpu_fuzz_plan.py runs the same rounds inside the suite)"""
import math
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from path_planner_amd import workloads
from path_planner_amd.types import make_config, H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K
from test_gpu_host_planner import _write_map, _scenario, _run_cli, _compare, CLI
import oracle as orc
DEVICES = [int(x) for x in os.environ["FUZZ_DEVICES"].split()] if os.environ.get("FUZZ_DEVICES") else None   # e.g. "0 0": two contexts (round trips in flight) on device 0


def judge(host, st, plan):
    for k, v in (("samples", st.samples), ("iterations", st.iterations), ("first_goal_iteration", st.first_goal_iteration)):
        if host[k] != v:
            return f"{k}: host {host[k]} oracle {v}"
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1.0)
    if rel(host["plan_f"], st.plan_f) > 1e-5:
        return f"plan_f: host {host['plan_f']} oracle {st.plan_f}"
    for k, v in (("expanded", st.expanded), ("generated", st.generated)):
        if host[k] != v:
            # a vertex pruned on one side only (`best.f < v.f`, SamplingBasedPlanner.cpp:11) moves these counters by a few; like
            # another plan of equal cost it has to be explained by the edge dumps
            return f"tie ({k}: host {host[k]} oracle {v})"
    try:
        _compare(host, st, plan, allow_order_fallbacks=True)
        return "ok"
    except AssertionError as ex:
        if os.environ.get("FUZZ_VERBOSE"):
            import traceback
            print("   _compare:", traceback.format_exc().strip().splitlines()[-3:], flush=True)
        return "tie"


def edge_dumps(w, sc, mp, t0, dt, calls, init, world, prev=None, start=None):
    """Every costed edge each search consumes, in consumption order, 16 numbers per edge: source state (5), Dubins parameters
    (3), word, radius, coverage flag, infeasible, true cost, g, h, end time.  Host: PPAMD_DUMP_EDGES; oracle: dump_edges."""
    _scenario(w, sc, mp, t0, dt, calls, init, speculation=1, prev=prev, start=start, devices=DEVICES)
    dump = sc + ".edges"
    subprocess.run([CLI, sc], capture_output=True, text=True, timeout=300, env=dict(os.environ, PPAMD_DUMP_EDGES=dump))
    H = np.loadtxt(dump).reshape(-1, 16)
    rc, st, plan, _, O = world.plan(w.ribbons4, w.start5 if start is None else start, calls * dt, t0, dt, initial_samples=init,
                                    dump_edges=200000, prev11=prev)
    return H, O


def degenerate_dubins(h, o):
    """Two different curves from the same source: is the Dubins problem one whose answer hangs on the last bits of its input?
    The target is read off the oracle's curve (its end pose); the problem is solved again (oracle's solver, glibc) with source
    and target moved by 1e-13 .. 1e-11 in heading and position.  If some such perturbation returns the HOST's curve (same word,
    length within 1e-6) the choice between the two is decided below the accuracy of any libm: two words on the edge of
    existing, or angles of +-1e-16 that `mod2pi` turns into 0 or a full turn (kinds i and ii)."""
    rho = float(o[9])
    if float(h[9]) != rho:
        return False
    q0 = [float(o[0]), float(o[1]), orc.yaw(float(o[2]))]
    lo_len = float(o[5] + o[6] + o[7]) * rho
    e, q1 = orc.dubins_sample([q0[0], q0[1], q0[2], float(o[5]), float(o[6]), float(o[7]), rho, float(o[8])], lo_len)
    if e != 0:
        e, q1 = orc.dubins_sample([q0[0], q0[1], q0[2], float(o[5]), float(o[6]), float(o[7]), rho, float(o[8])], lo_len - 1e-9)
        if e != 0:
            return False
    want_word, want_len = float(h[8]), float(h[5] + h[6] + h[7]) * rho
    return orc.dubins_answer_hangs_on_last_bits(q0, q1, rho, want_word, want_len)


def full_turns_apart(h, o):
    """The same word with the same parameters up to whole turns of an arc: an angle of +-1e-16 that `mod2pi` made 0 on one side
    and 2 pi on the other (collinear poses; DESIGN.md Appendix C kind ii)."""
    if h[8] != o[8] or h[9] != o[9]:
        return False
    for i in range(3):
        d = abs(float(h[5 + i]) - float(o[5 + i]))
        straight = i == 1 and h[8] < 4
        if not (d < 1e-9 or (not straight and abs(d - 2 * math.pi) < 1e-9)):
            return False
    return True


def end_pose(row):
    q0 = [float(row[0]), float(row[1]), orc.yaw(float(row[2]))]
    ln = float(row[5] + row[6] + row[7]) * float(row[9])
    p8 = [q0[0], q0[1], q0[2], float(row[5]), float(row[6]), float(row[7]), float(row[9]), float(row[8])]
    e, q = orc.dubins_sample(p8, ln)
    if e != 0:
        e, q = orc.dubins_sample(p8, max(ln - 1e-9, 0.0))
    return q


def same_target(h, o):
    a, b = end_pose(h), end_pose(o)
    dth = abs(a[2] - b[2]) % (2 * math.pi)
    return math.hypot(a[0] - b[0], a[1] - b[1]) < 1e-4 and min(dth, 2 * math.pi - dth) < 1e-4


def classify_tie(H, O, same_length=True):
    """Is a 'tie' verdict one of the two explained kinds?  Returns (True, what) or (False, why not)."""
    if same_length and len(H) != len(O):
        return False, f"the searches consume {len(H)} and {len(O)} edges"
    rel = lambda a, b: np.abs(a - b) / np.maximum(1.0, np.abs(b))
    word_tie = retry = degenerate = lastbit = False
    for i in range(min(len(H), len(O))):
        h, o = H[i], O[i]
        if np.max(rel(h[:5], o[:5])) > 1e-4:
            break       # from here on the two searches expand different (equal-f) vertices: only what came before can explain it
        if h[8] != o[8]:
            # same source, two words for (nearly) the same length: a zero-length arc makes two words the same curve; otherwise the
            # lengths must agree to the last digits (a tie `cost < best` settles by rounding), or to 1e-5 once an upstream end
            # pose has moved
            lh, lo = (h[5] + h[6] + h[7]) * h[9], (o[5] + o[6] + o[7]) * o[9]
            same_curve = min(h[5], h[7]) <= 1e-9 and min(o[5], o[7]) <= 1e-9
            if h[9] != o[9] or (not same_curve and abs(lh - lo) > (1e-5 if retry else 1e-11) * max(1.0, abs(lo))):
                if not same_target(h, o):
                    break                                   # same source, another target: the order of equal-f siblings already differs
                if degenerate_dubins(h, o):
                    degenerate = True
                    continue
                return False, f"edge {i}: different Dubins words, lengths {lh!r} and {lo!r}"
            word_tie = True
            continue
        r = rel(h, o)
        if np.max(r) > 1e-4:
            if np.max(rel(h[5:8], o[5:8])) > 1e-4 and (full_turns_apart(h, o) or (same_target(h, o) and degenerate_dubins(h, o))):
                degenerate = True                           # the same word with an arc of 0 on one side and a full turn on the other
                continue
            break       # same source, another target: the order of equal-f siblings already differs
        if h[11] != o[11]:
            return False, f"edge {i}: infeasible flag differs"
        if np.max(r) > 1e-12:
            retry = True                                    # same edge, last digits differ: a parent's end pose moved by <= 1e-5 m
        elif np.any(h != o):
            lastbit = True                                  # same edge to the last bit or two: h from child ribbons that carry the
                                                            # sweeps' own rounding (corridor runs, per-step sincos: DESIGN.md Appendix C)
    if not (word_tie or retry or degenerate or lastbit):
        return False, "no upstream difference explains the other plan: the push / pop order itself differs"
    return True, " + ".join(x for x, on in (("equal-length Dubins words", word_tie), ("one-sided 1e-5 retry upstream", retry),
                                            ("a Dubins problem whose shortest word flips under a 1e-11 perturbation", degenerate),
                                            ("costs that differ in the last bit upstream (f ties broken by an ulp of h)", lastbit and not (word_tie or retry or degenerate))) if on)


def make_round(rng, rid):
    size = int(rng.choice([256, 512])); res = float(rng.choice([0.25, 0.5]))
    ext = size * res; c = ext / 2
    grid = np.zeros((size, size), dtype=np.uint8)
    for _ in range(int(rng.integers(0, 10))):
        a, b = rng.integers(0, size - 24, 2)
        grid[a:a + rng.integers(4, 24), b:b + rng.integers(4, 24)] = 1
    i0 = int(c / res); grid[i0 - 16:i0 + 16, i0 - 16:i0 + 16] = 0
    heur = int(rng.choice([H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K, H_TSP_POINT_K]))
    nrib = int(rng.integers(1, 5))
    rib = []
    for i in range(nrib):
        x, y = rng.uniform(0.3 * ext, 0.7 * ext, 2); L, th = rng.uniform(8, 0.25 * ext), rng.uniform(0, 2 * np.pi)
        rib.append([x, y, min(max(x + L * np.cos(th), 2), ext - 2), min(max(y + L * np.sin(th), 2), ext - 2)])
    max_speed = float(rng.choice([2.5, 4.0]))
    cfg = make_config(start_state_time=1.0, heuristic=heur, tsp_k=int(rng.integers(1, 4)), max_speed=max_speed,
                      slow_speed=float(rng.choice([-1.0, 0.5])), turning_radius=float(rng.choice([4.0, 8.0])),
                      coverage_turning_radius=float(rng.choice([8.0, 16.0])), time_horizon=float(rng.choice([15.0, 30.0])),
                      time_minimum=float(rng.choice([2.0, 5.0])), collision_checking_increment=float(rng.choice([0.05, 0.11, 0.25])),
                      ribbon_width=float(rng.choice([1.0, 1.5, 3.0])))
    nob = int(rng.integers(0, 12))
    ob = workloads.obstacles(nob, int(rng.integers(1, 1 << 30)), ext, time=1.0, keep_free=(c, c, 10)) if nob else None
    w = workloads.Workload(f"fuzz{rid}", grid, res, ob, rib, [c, c, float(rng.uniform(0, 6.28)), max_speed, 1.0], 0, 7, cfg)
    t0, dt = float(rng.choice([1000.0, 1.6e9])), 1e-3
    calls, init = int(rng.integers(15, 60)), int(rng.choice([64, 256, 512]))
    spec = int(rng.choice([1, 16, 16]))
    tag = f"round {rid}: grid {size}@{res} rib {nrib} heur {heur} obst {nob} calls {calls} init {init}"
    return w, t0, dt, calls, init, spec, tag


def one_round(rng, rid, d, verbose=True):
    """-> list of (which, verdict, explanation) for the first plan() call and, when it returned a plan, the replan."""
    w, t0, dt, calls, init, spec, tag = make_round(rng, rid)
    cfg, grid, res, ob = w.cfg, w.grid, w.res, w.obst
    orc.O.ppo_set_ribbon_width(cfg.ribbon_width)
    world = orc.World(cfg, grid, res, ob)
    mp = os.path.join(d, "grid.map"); _write_map(grid, res, mp)
    sc = os.path.join(d, "s.txt")
    say = (lambda *a: print(*a, flush=True)) if verbose else (lambda *a: None)
    out = []

    def run(which, t_start, prev, start):
        _scenario(w, sc, mp, t_start, dt, calls, init, speculation=spec, prev=prev, start=start, devices=DEVICES)
        host = _run_cli(sc)
        rc, st, plan, _, _ = world.plan(w.ribbons4, w.start5 if start is None else start, calls * dt, t_start, dt, initial_samples=init, prev11=prev)
        if rc != 0 or "exception" in host:
            same = (rc != 0) == ("exception" in host)
            out.append((which, "both threw" if same else "MISMATCH", "" if same else f"one side threw (oracle rc {rc}, host {host.get('exception')})"))
            return host, st, None
        v = judge(host, st, plan)
        why = ""
        if v == "ok" and host.get("order_fallbacks", 0):
            why = f"same plan; {host['order_fallbacks']} lists pushed in ascending length (not replayable: see _compare)"
        if v.startswith("tie"):
            counts = v[3:].strip()
            ok, why = classify_tie(*edge_dumps(w, sc, mp, t_start, dt, calls, init, world, prev=prev, start=start), same_length=not counts)
            why = (counts + " " + why).strip()
            v = "tie" if ok else "MISMATCH"
        elif v.startswith("plan_f:"):
            # another plan of ANOTHER cost: inside a budget of clock calls two searches that part at an explained tie go on to expand
            # different vertices and may end with different incumbents.  Accepted only when the edge dumps agree up to a divergence of
            # one of the explained kinds (classify_tie); reported with both costs
            ok, why2 = classify_tie(*edge_dumps(w, sc, mp, t_start, dt, calls, init, world, prev=prev, start=start), same_length=False)
            why = v + " | " + why2
            v = "tie" if ok else "MISMATCH"
        elif v != "ok":
            why, v = v, "MISMATCH"
        out.append((which, v, why))
        if os.environ.get("FUZZ_VERBOSE") and v != "ok":
            hp = np.array(host["plan"], dtype=np.float64).reshape(-1, 11)
            print(which, v, why, "| depth", host.get("plan_depth"), st.plan_depth, "plan_f", repr(host["plan_f"]), repr(st.plan_f), "plan_h", repr(host.get("plan_h")), repr(st.plan_h))
            for i in range(max(len(hp), len(plan))):
                if i < len(hp): print("  H", i, np.array2string(hp[i], precision=15, max_line_width=400))
                if i < len(plan): print("  O", i, np.array2string(plan[i], precision=15, max_line_width=400))
        return host, st, plan

    host, st, plan = run("plan", t0, None, None)
    if plan is not None and out[-1][1] in ("ok", "tie") and len(plan):
        seg = plan[0]
        e, q = orc.dubins_sample(seg[:8], min(1.0 * seg[8], (seg[10] - seg[9]) * seg[8]))
        hdg = math.pi / 2 - q[2]
        hdg += 2 * math.pi if hdg < 0 else 0
        start2 = np.array([q[0], q[1], hdg, seg[8], seg[9] + 1.0])
        if start2[4] < plan[-1][10] - 1e-6:
            cfg.start_state_time = float(start2[4]); world.set_config(cfg)
            run("replan", t0 + 1.0, plan, start2)
            cfg.start_state_time = 1.0
    for which, v, why in out:
        say(tag if which == "plan" else "    replan", "->", v, ("(" + why + ")") if why else "",
            ("expanded %d first goal %d f %.4f" % (host["expanded"], host["first_goal_iteration"], host["plan_f"])) if which == "plan" and "expanded" in host else "")
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    only = {int(x) for x in sys.argv[3:]}          # optional: replay just these rounds of the sequence
    bad = ties = calls = 0
    with tempfile.TemporaryDirectory() as d:
        for r in range(rounds):
            if only and r not in only:
                make_round(rng, r)                 # advance the generator as the round would have
                continue
            for which, v, why in one_round(rng, r, d):
                calls += 1
                bad += v == "MISMATCH"
                ties += v == "tie"
    orc.O.ppo_set_ribbon_width(1.5)
    print(f"{rounds} rounds, {calls} plan() calls: {bad} mismatches, {ties} equal-cost plans explained by the edge dumps")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
