#!/bin/bash
# GPU box: WRITE_SIZE (HBM bytes written, own PMC pass) per costing kernel for variants of the library side by side.
# usage: tools/write_size_ab.sh ["name=flags" ...]    default: fuse0=-DPP_FUSE_HEUR=0 and the default build
set -e
mkdir -p gpurun_out/abl
variants=("$@"); [ ${#variants[@]} -eq 0 ] && variants=("fuse0=-DPP_FUSE_HEUR=0")
variants+=("default=")
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
for v in "${variants[@]}"; do
  name=${v%%=*}; flags=${v#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags path_planner_amd/csrc/ppgpu.hip -o gpurun_out/abl/libppgpu_$name.so -ldl
  export PPGPU_LIB_OVERRIDE=$PWD/gpurun_out/abl/libppgpu_$name.so
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/wr_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/wr_$name.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/wr_$name/*/*counter_collection.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE' and ('sweep' in r['Kernel_Name'] or 'heuristic(' in r['Kernel_Name']):
        d[r['Kernel_Name'][:18]].append(float(r['Counter_Value']))
print('$name', {k: round(sum(v) / len(v) * 1024 / 1e6, 1) for k, v in d.items()}, 'MB written per launch')
PY
done
