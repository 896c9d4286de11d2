#!/bin/bash
# GPU box: where a 100 ms plan() goes.  usage: tools/prof_plan.sh [rocprof]   (host-side laps always; HIP API / kernel stats with "rocprof")
set -e
python3 tools/make_scenario.py /tmp/sc "time_remaining 0.1" "real_clock 1" "repeat 6" > /dev/null
mkdir -p gpurun_out
PPAMD_PROFILE=1 path_planner_amd/host/plan_cli /tmp/sc/s.txt 2> gpurun_out/planlaps.txt | tail -c 400 | cut -c1-300
cat gpurun_out/planlaps.txt
if [ "$1" == "rocprof" ]; then
  cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/planprof -o pp -- path_planner_amd/host/plan_cli /tmp/sc/s.txt > gpurun_out/planprof.log 2>&1
  head -8 gpurun_out/planprof/pp_hip_api_stats.csv
  head -14 gpurun_out/planprof/pp_kernel_stats.csv | cut -c1-110
fi
