#!/bin/bash
# GPU box: the counter-anchored utilisation of the costing kernels -> gpurun_out/valu.json (copied to
# profiles/valu.json, which bench.py reads for roofline.valu_issue_frac).  Two PMC passes over three bench steps, counters only:
#   SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES   wave-instructions and the quad-cycles the VALU was issuing
#   GRBM_GUI_ACTIVE                                            shader cycles of each dispatch (summed over the 8 XCDs)
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PP_BENCH_PROFILED=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/valu_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/valu_sq.log 2>&1 || { tail -5 gpurun_out/valu_sq.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/valu_clk -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/valu_clk.log 2>&1 || { tail -5 gpurun_out/valu_clk.log; exit 1; }
python3 tools/valu_report.py
