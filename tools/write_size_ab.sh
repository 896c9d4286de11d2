set -e
mkdir -p gpurun_out/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DPP_FUSE_HEUR=0 path_planner_amd/csrc/ppgpu.hip -o gpurun_out/abl/libppgpu_fuse0.so -ldl
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in fuse0 default; do
  if [ $v == fuse0 ]; then export PPGPU_LIB_OVERRIDE=$PWD/gpurun_out/abl/libppgpu_fuse0.so; else unset PPGPU_LIB_OVERRIDE; fi
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/wr_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/wr_$v.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/wr_$v/*/*counter_collection.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE' and ('sweep' in r['Kernel_Name'] or 'heuristic' in r['Kernel_Name'] or 'solve' in r['Kernel_Name']):
        d[r['Kernel_Name'][:18]].append(float(r['Counter_Value']))
for k, v in d.items(): print('$v', k, [round(x * 1024 / 1e6, 1) for x in v])
PY
done
