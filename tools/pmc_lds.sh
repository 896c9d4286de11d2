#!/bin/bash
# GPU box: LDS-side counters of one bench step (own pass: counters only with --kernel-trace)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PP_BENCH_PROFILED=1      # bench.py under rocprofv3: only the headline launches (no open-vertex run, no plan()-level legs)
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_lds.log 2>&1 || { tail -5 gpurun_out/pmc_lds.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_lds/*/*counter_collection.csv')[0]
d = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    d[(r['Kernel_Name'][:24], r['Counter_Name'])] += float(r['Counter_Value'])
for k in sorted({k[0] for k in d}):
    if k.startswith('pp_k_') and d[(k, 'SQ_INSTS_VALU')] > 1e6:
        print(k, {c: d[(k, c)] for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_LDS')})
PY
