#!/bin/bash
# GPU box: the evidence of one round in one go.  usage: tools/round_profile.sh <tag>   (writes gpurun_out/<tag>_*)
set -e
tag=$1
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1
python3 bench.py > gpurun_out/${tag}_bench.log 2>&1
grep '^{' gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o ${tag} --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
grep '^{' gpurun_out/${tag}_prof.log > gpurun_out/${tag}_bench_under_rocprof.json
bash tools/traffic.sh > gpurun_out/${tag}_traffic.log 2>&1
echo done
