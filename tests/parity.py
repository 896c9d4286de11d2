"""Shared parity checker: HIP results vs the CPU oracle's for the same edges.

Bar (BASELINE.json north_star): flags and integer fields identical; costs within 1e-5 relative.
The tolerance used for every floating field is REL_TOL below (abs floor 1e-9 for values near 0).
"""
import numpy as np

from path_planner_amd.types import F_INFEASIBLE, F_THROWS

REL_TOL = 1e-5
ABS_FLOOR = 1e-9
FLOAT_FIELDS = ["true_cost", "collision_penalty", "approx_cost", "end_x", "end_y", "end_heading", "end_speed", "end_time",
                "g", "h", "f", "coverage_completed_time"]


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return d / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1.0)


def compare_results(gpu, cpu, gpu_child=None, cpu_child=None, allow_word_ties=False, skip_heuristic=False):
    """allow_word_ties: two Dubins words can have (numerically) the same length — the same curve with a zero-length last arc, or
    mirror-image curves when source and target are symmetric about the line joining them; which of them `cost < best`
    keeps is decided by the last bit of the libm in use, on the reference as well.  With this option an edge whose word
    differs but whose path length agrees to 1e-12 is counted (n_word_ties) and excluded from the comparison.  skip_heuristic: do not compare h and f (the Dubins-TSP heuristics on collinear split pieces sit on
    the Dubins mod-2pi discontinuity).  Both are used by the randomized test's second generation
    only; every other test runs with the strict defaults."""
    rep = {"n": int(len(gpu))}
    # Dubins word, child ribbon count, executed steps
    throws = (cpu["flags"] & F_THROWS) != 0
    info_ok = (gpu["info"] == cpu["info"]) | throws
    tie = np.zeros(len(gpu), dtype=bool)
    if allow_word_ties:
        tie = ((gpu["info"] & 0xFF) != (cpu["info"] & 0xFF)) & ~throws & (_rel(gpu["approx_cost"], cpu["approx_cost"]) <= 1e-12)
        info_ok = info_ok | tie
    rep["flags_equal"] = bool(np.array_equal(gpu["flags"][~tie], cpu["flags"][~tie]))
    rep["n_flag_mismatch"] = int(np.count_nonzero((gpu["flags"] != cpu["flags"]) & ~tie))
    rep["n_word_ties"] = int(np.count_nonzero(tie))
    rep["n_info_mismatch"] = int(np.count_nonzero(~info_ok))
    feas = ((cpu["flags"] & F_INFEASIBLE) == 0) & ((gpu["flags"] & F_INFEASIBLE) == 0) & ~tie
    rep["n_feasible"] = int(np.count_nonzero(feas))
    worst = 0.0
    for f in FLOAT_FIELDS:
        if skip_heuristic and f in ("h", "f"):
            continue
        if np.any(feas):
            r = _rel(gpu[f][feas], cpu[f][feas])
            m = float(np.nanmax(r)) if r.size else 0.0
            if not np.array_equal(np.isnan(gpu[f][feas]), np.isnan(cpu[f][feas])):
                m = float("inf")
        else:
            m = 0.0
        rep["rel_" + f] = m
        worst = max(worst, m)
    if np.any(feas):
        r = _rel(gpu["param"][feas], cpu["param"][feas])
        rep["rel_param"] = float(r.max()) if r.size else 0.0
        worst = max(worst, rep["rel_param"])
        rep["bit_identical_cost_frac"] = float(np.mean(gpu["true_cost"][feas] == cpu["true_cost"][feas]))
    if gpu_child is not None and cpu_child is not None and np.any(feas):
        r = _rel(gpu_child[feas], cpu_child[feas])
        rep["rel_child_ribbons"] = float(r.max())
        worst = max(worst, rep["rel_child_ribbons"])
    rep["worst_rel"] = worst
    rep["ok"] = bool(rep["flags_equal"] and rep["n_info_mismatch"] == 0 and worst <= REL_TOL)
    return rep
