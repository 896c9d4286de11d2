#!/usr/bin/env python3
"""GPU box: randomized differential test of whole plan() calls — the C++ host planner (plan_cli, injected clock) against the
oracle's restatement of AStarPlanner::plan on random worlds: grid, obstacles, ribbons, heuristic, speeds, radii, budget, and
a second cycle that hands the first plan back (previous-plan re-costing).  Identical samples / iterations / expansions /
generated / first-goal iteration / depth, costs within 1e-5.  usage: tools/fuzz_plan.py [rounds] [seed]"""
import os, sys, tempfile, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from path_planner_amd import workloads
from path_planner_amd.types import make_config, H_MAX_DISTANCE, H_TSP_POINT_ALL, H_TSP_POINT_K
from test_gpu_host_planner import _write_map, _scenario, _run_cli, _compare
import oracle as orc


def judge(host, st, plan):
    """'ok' / 'tie' (same statistics and cost, another plan among vertices of exactly equal f: DESIGN.md 4.5) / the failed check"""
    for k, v in (("samples", st.samples), ("iterations", st.iterations), ("expanded", st.expanded), ("generated", st.generated),
                 ("first_goal_iteration", st.first_goal_iteration)):
        if host[k] != v:
            return f"{k}: host {host[k]} oracle {v}"
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b), 1.0)
    if rel(host["plan_f"], st.plan_f) > 1e-5:
        return f"plan_f: host {host['plan_f']} oracle {st.plan_f}"
    try:
        _compare(host, st, plan)
        return "ok"
    except AssertionError:
        return "tie"


def diff_edges(w, sc, mp, t0, dt, calls, init, world):
    """Which costed edges differ?  Host (PPAMD_DUMP_EDGES) and oracle (dump_edges) list every edge they cost; within one
    expansion the two order siblings differently, so the lists are compared as multisets keyed by source time, word, radius and
    rounded parameters."""
    import subprocess
    from test_gpu_host_planner import CLI
    _scenario(w, sc, mp, t0, dt, calls, init, speculation=1)
    dump = sc + ".edges"
    subprocess.run([CLI, sc], capture_output=True, text=True, timeout=300, env=dict(os.environ, PPAMD_DUMP_EDGES=dump))
    H = np.loadtxt(dump).reshape(-1, 16)
    rc, st, plan, _, O = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init, dump_edges=200000)
    key = lambda r: (round(r[4], 6), int(r[8]), r[9], int(r[10]), round(r[5], 5), round(r[6], 5), round(r[7], 5))
    hk, ok_ = {}, {}
    for r in H: hk.setdefault(key(r), []).append(r)
    for r in O: ok_.setdefault(key(r), []).append(r)
    only_h = [k for k in hk if k not in ok_]; only_o = [k for k in ok_ if k not in hk]
    flips = [k for k in hk if k in ok_ and hk[k][0][11] != ok_[k][0][11]]
    print(f"      edges costed: host {len(H)} oracle {len(O)}; only host {len(only_h)}, only oracle {len(only_o)}, infeasible flag differs on {len(flips)}", flush=True)
    for k in (flips[:3] + only_o[:3] + only_h[:3]):
        print("        ", k, "host", [list(np.round(r[11:16], 9)) for r in hk.get(k, [])][:1], "oracle", [list(np.round(r[11:16], 9)) for r in ok_.get(k, [])][:1], flush=True)


def one_round(rng, rid, d):
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
    orc.O.ppo_set_ribbon_width(cfg.ribbon_width)
    world = orc.World(cfg, grid, res, ob)
    t0, dt = float(rng.choice([1000.0, 1.6e9])), 1e-3
    calls, init = int(rng.integers(15, 60)), int(rng.choice([64, 256, 512]))
    mp = os.path.join(d, "grid.map"); _write_map(grid, res, mp)
    sc = os.path.join(d, "s.txt")
    _scenario(w, sc, mp, t0, dt, calls, init, speculation=int(rng.choice([1, 16, 16])))
    host = _run_cli(sc)
    rc, st, plan, _, _ = world.plan(w.ribbons4, w.start5, calls * dt, t0, dt, initial_samples=init)
    tag = f"round {rid}: grid {size}@{res} rib {nrib} heur {heur} obst {nob} calls {calls} init {init}"
    if rc != 0 or "exception" in host:
        same = (rc != 0) == ("exception" in host)
        print(tag, "-> both threw" if same else f"-> ONE SIDE THREW (oracle rc {rc}, host {host.get('exception')})", flush=True)
        return same
    verdict = judge(host, st, plan)
    if verdict not in ("ok", "tie"):
        print(tag, "-> MISMATCH", verdict, {k: host[k] for k in host if k != "plan"}, flush=True)
        diff_edges(w, sc, mp, t0, dt, calls, init, world)
        for spec in (1, 4, 16):                       # does the host's own answer depend on how it batches?
            _scenario(w, sc, mp, t0, dt, calls, init, speculation=spec)
            h2 = _run_cli(sc)
            print("      speculation", spec, {k: h2[k] for k in ("samples", "iterations", "expanded", "generated", "first_goal_iteration", "plan_f")}, flush=True)
        return False
    ok2 = True
    if len(plan):
        seg = plan[0]
        e, q = orc.dubins_sample(seg[:8], min(1.0 * seg[8], (seg[10] - seg[9]) * seg[8]))
        hdg = math.pi / 2 - q[2]
        hdg += 2 * math.pi if hdg < 0 else 0
        start2 = np.array([q[0], q[1], hdg, seg[8], seg[9] + 1.0])
        if start2[4] < plan[-1][10] - 1e-6:
            sc2 = os.path.join(d, "s2.txt")
            _scenario(w, sc2, mp, t0 + 1.0, dt, calls, init, prev=plan, start=start2)
            host2 = _run_cli(sc2)
            cfg.start_state_time = float(start2[4]); world.set_config(cfg)
            rc2, st2, plan2, _, _ = world.plan(w.ribbons4, start2, calls * dt, t0 + 1.0, dt, initial_samples=init, prev11=plan)
            if rc2 != 0 or "exception" in host2:
                ok2 = (rc2 != 0) == ("exception" in host2)
                print("    replan:", "both threw" if ok2 else f"ONE SIDE THREW (oracle rc {rc2}, host {host2.get('exception')})", flush=True)
            else:
                v2 = judge(host2, st2, plan2)
                if v2 not in ("ok", "tie"):
                    ok2 = False
                    print("    replan MISMATCH", v2, {k: host2[k] for k in host2 if k != "plan"}, flush=True)
                elif v2 == "tie":
                    print("    replan: equal-cost plan among f ties", flush=True)
    print(tag, "-> " + ("ok" if verdict == "ok" else "equal-cost plan among f ties"), "expanded", host["expanded"], "first goal", host["first_goal_iteration"], "f %.4f" % host["plan_f"], flush=True)
    return ok2


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for r in range(rounds):
            bad += 0 if one_round(rng, r, d) else 1
    orc.O.ppo_set_ribbon_width(1.5)
    print(f"{rounds} rounds, {bad} with mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
