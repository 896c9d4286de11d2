#!/bin/bash
# GPU box: does the records' D2H copy of bench.py's end-to-end loops run as a blit KERNEL (__amd_rocclr_copyBuffer, on the CUs) or on an
# SDMA engine, and what do e2e_ms_per_step / e2e_pipelined_ms_per_step become either way?  
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1
mkdir -p gpurun_out
run() {
    name=$1; shift
    rm -rf gpurun_out/ce_$name
    ( export "$@" X_UNUSED=1; timeout -k 10 150 rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/ce_$name -o c --output-format csv -- python3 bench.py --no-cpu-baseline --no-plan-level --steps 10 --warmup 3 > gpurun_out/ce_$name.log 2>&1 )
    echo "== $name ($*)"
    python3 - "$name" <<'PY'
import csv, json, sys, os
name = sys.argv[1]
for l in open(f"gpurun_out/ce_{name}.log"):
    if l.startswith("{"):
        d = json.loads(l); print("   ms_per_step %.4f  e2e %.4f  e2e_pipelined %.4f" % (d["ms_per_step"], d["e2e_ms_per_step"], d["e2e_pipelined_ms_per_step"]))
ks = f"gpurun_out/ce_{name}/c_kernel_stats.csv"
for r in csv.DictReader(open(ks)):
    if "copyBuffer" in r["Name"] or "fillBuffer" in r["Name"]:
        print(f"   kernel {r['Name'][:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f} us")
ms = f"gpurun_out/ce_{name}/c_memory_copy_stats.csv"
if os.path.exists(ms):
    for r in csv.DictReader(open(ms)):
        print(f"   copy   {r['Name'][:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f} us")
PY
}
run default PP_X=0
run sdma_engine GPU_BLIT_ENGINE_TYPE=2
run hsa_sdma1 HSA_ENABLE_SDMA=1
run hsa_sdma0 HSA_ENABLE_SDMA=0
