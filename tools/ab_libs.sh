#!/bin/bash
# GPU box: A/B of prebuilt libraries build/libppgpu_<name>.so on one box: rocprofv3 kernel averages of a short bench run, twice each, and
# the records hash.  usage: tools/ab_libs.sh nameA nameB
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1
for n in $1 $2 $1 $2; do
  export PPGPU_LIB_OVERRIDE=$PWD/build/libppgpu_$n.so
  rm -rf gpurun_out/ab_$n
  rocprofv3 --kernel-trace --stats -d gpurun_out/ab_$n -o v --output-format csv -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab_$n.log 2>&1
  python3 - $n <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(f"gpurun_out/ab_{sys.argv[1]}/v_kernel_stats.csv")))
d={r['Name'].split('(')[0]:float(r['AverageNs'])/1e3 for r in rows}
ms=[json.loads(l)['ms_per_step'] for l in open(f"gpurun_out/ab_{sys.argv[1]}.log") if l.startswith('{')]
print(sys.argv[1], {k:round(d[k],1) for k in ('pp_k_cover_sweep','pp_k_pose_sweep','pp_k_plan_skips','pp_k_approach_events','pp_k_cover_finish')}, 'ms_per_step', ms)
PY
done
for n in $1 $2; do PPGPU_LIB_OVERRIDE=$PWD/build/libppgpu_$n.so python3 tools/records_hash.py | grep -v amdgpu; done
