#!/bin/bash
# GPU box: build variants of the library (name=flags ...) and print rocprofv3 kernel averages of a short bench run for each.
# usage: tools/variant_stats.sh name1 "flags1" [name2 "flags2" ...]
mkdir -p gpurun_out/abl
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags path_planner_amd/csrc/ppgpu.hip -o gpurun_out/abl/libppgpu_$name.so -ldl || exit 1
    export PPGPU_LIB_OVERRIDE=$PWD/gpurun_out/abl/libppgpu_$name.so
    rm -rf gpurun_out/vs_$name
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/vs_$name -o v --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/vs_$name.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "$name: rc $rc"; tail -5 gpurun_out/vs_$name.log; exit $rc; fi
    echo "== $name ($flags)"
    python3 - "$name" <<'PY'
import csv, sys, json
name = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/vs_{name}/v_kernel_stats.csv")))
tot = 0.0
for r in rows:
    if r["Name"].startswith("pp_k_") and int(r["Calls"]) >= 13 and float(r["AverageNs"]) > 8000 and not any(x in r["Name"] for x in ("chain", "compact", "proj", "lengths")):
        print(f"   {r['Name'][:34]:36s} {float(r['AverageNs'])/1e3:9.1f} us"); tot += float(r["AverageNs"])
print(f"   sum {tot/1e3:.1f} us")
for l in open(f"gpurun_out/vs_{name}.log"):
    if l.startswith("{"):
        d = json.loads(l); print("   ms_per_step", round(d["ms_per_step"], 4), "parity", d.get("parity", {}).get("ok"))
PY
done
