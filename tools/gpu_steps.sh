#!/bin/bash
# Run GPU steps one after another on the box, each under its own time limit; a step that FAILS lets the next one run, a step that
# was KILLED at its limit (or died on a signal) stops the sequence (no further GPU work after a hang).  Output of step i goes to
# gpurun_out/<tag>_<i>.log.   usage: tools/gpu_steps.sh <tag> <seconds> '<cmd>' [<seconds> '<cmd>' ...]
tag=$1; shift
mkdir -p gpurun_out
i=0
while [ $# -ge 2 ]; do
    lim=$1; cmd=$2; shift 2
    i=$((i+1))
    echo "== step $i (limit ${lim}s): $cmd"
    timeout -k 10 "$lim" bash -c "$cmd" > "gpurun_out/${tag}_${i}.log" 2>&1
    rc=$?
    echo "== step $i rc=$rc"; tail -n 6 "gpurun_out/${tag}_${i}.log"
    if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "== step $i was killed: stopping"; exit $rc; fi
done
exit 0
