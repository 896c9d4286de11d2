#!/usr/bin/env python3
"""GPU box, developer aid: why did pp_k_expand_order fall back in a round of tools/fuzz_plan.py?  Needs the library built with
-DPP_DBG_ORD in place (the C++ planner links path_planner_amd/csrc/libppgpu.so):
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DPP_DBG_ORD path_planner_amd/csrc/ppgpu.hip -o path_planner_amd/csrc/libppgpu.so -ldl
usage: tools/dbg_order_fallback.py <seed> <round>"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import fuzz_plan as fp
import test_gpu_host_planner as T
orig = T._run_cli


def run(sc):
    r = subprocess.run([T.CLI, sc], capture_output=True, text=True, timeout=300)
    fb = [l for l in r.stdout.splitlines() + r.stderr.splitlines() if "FALLBACK" in l]
    if fb:
        print("\n".join(fb[:8]), flush=True)
    return orig(sc)


fp._run_cli = run
seed, rid = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for r in range(rid):
    fp.make_round(rng, r)
with tempfile.TemporaryDirectory() as d:
    print(fp.one_round(rng, rid, d, verbose=False))
