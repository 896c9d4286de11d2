import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "gpurun_out", "libppgpu_dbgord.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-DPP_DBG_ORD",
                       os.path.join(ROOT, "path_planner_amd", "csrc", "ppgpu.hip"), "-o", lib, "-ldl"])
env = dict(os.environ, PPGPU_LIB_OVERRIDE=lib)
for n in sys.argv[1:] or ["16384", "262144"]:
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_expand_order.py"), n], env=env, capture_output=True, text=True)
    print("\n".join([l for l in out.stdout.splitlines() if l.startswith("[ord]")][-8:]))
    print(out.stdout.splitlines()[-1])
