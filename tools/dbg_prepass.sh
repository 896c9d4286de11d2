#!/bin/bash
# GPU box, developer aid: which prepass makes tools/dbg_lane_h.py's case differ?  Builds variants and runs the case with each.
mkdir -p gpurun_out/abl
for v in "noappr:-DPP_NO_APPROACH" "noskip:-DPP_NO_CHUNK_SKIP" "nolane:-DPP_LANE_HEUR=0"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 $flags path_planner_amd/csrc/ppgpu.hip -o gpurun_out/abl/libppgpu_$name.so -ldl || exit 1
    echo "== $name"
    PPGPU_LIB_OVERRIDE=$PWD/gpurun_out/abl/libppgpu_$name.so python3 tools/dbg_lane_h.py "$@" 2>&1 | grep "differing\|edges whose"
done
