#!/bin/bash
# GPU box: where a 100 ms cycle of the config-5 replan loop goes.  usage: tools/prof_replan.sh [cycles] [initial samples] ["device ids", default "0 0" = two streams on device 0]
#   host-side laps of every plan() (PPAMD_PROFILE=1) -> gpurun_out/replan_laps.txt
#   rocprofv3 kernel / HIP API / copy statistics of the same loop -> gpurun_out/replanprof/
cycles=${1:-20}; init=${2:-8192}
mkdir -p gpurun_out
python3 tools/make_scenario.py /tmp/sc5 --cfg5 --initial $init "time_remaining 0.1" "replan $cycles 0.1" "devices ${3:-0 0}" > /dev/null || exit 1
PPAMD_PROFILE=1 path_planner_amd/host/plan_cli /tmp/sc5/s.txt 2> gpurun_out/replan_laps.txt | cut -c1-600
tail -n 8 gpurun_out/replan_laps.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/replanprof -o rp -- path_planner_amd/host/plan_cli /tmp/sc5/s.txt > gpurun_out/replanprof.log 2>&1
head -12 gpurun_out/replanprof/rp_hip_api_stats.csv | cut -c1-140
head -40 gpurun_out/replanprof/rp_kernel_stats.csv | cut -c1-140
