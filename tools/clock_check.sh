#!/bin/bash
# GPU box: shader clock under the bench's load = GRBM_GUI_ACTIVE cycles of a kernel / its duration (own PMC pass)
# usage: tools/clock_check.sh   -> gpurun_out/clock.txt
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
mkdir -p gpurun_out
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/clock -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/clock.log 2>&1
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob('gpurun_out/clock/*/*counter_collection.csv')[0]
kt = glob.glob('gpurun_out/clock/*/*kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE' or r['Dispatch_Id'] not in dur:
        continue
    name, ns = dur[r['Dispatch_Id']]
    if not any(k in name for k in ('pose_sweep', 'cover_sweep', 'k_heuristic')) or ns < 100000:
        continue
    a = acc[name[:20]]
    a[0] += float(r['Counter_Value']); a[1] += ns; a[2] += 1
with open('gpurun_out/clock.txt', 'w') as f:
    for k, (cyc, ns, n) in acc.items():
        line = f"{k:22s} launches {n}  GRBM_GUI_ACTIVE/launch {cyc / n:12.0f}  ns/launch {ns / n:10.0f}  => {cyc / ns:5.3f} GHz"
        print(line); f.write(line + "\n")
PY
