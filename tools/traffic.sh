#!/bin/bash
# GPU box: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950) over one bench
# step, then tools/traffic_report.py turns them into profiles/traffic.json.  Counters only: no sys/hip/hsa tracing.
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/traffic_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/traffic_$c.log 2>&1
done
python3 tools/traffic_report.py
