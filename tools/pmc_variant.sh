#!/bin/bash
# usage: tools/pmc_variant.sh name "flags" ; builds a variant library and collects SQ counters for one bench step
set -e
name=$1; flags=$2
mkdir -p gpurun_out/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags path_planner_amd/csrc/ppgpu.hip -o gpurun_out/abl/libppgpu_$name.so -ldl
export PPGPU_LIB_OVERRIDE=$PWD/gpurun_out/abl/libppgpu_$name.so
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
