#!/bin/bash
# GPU box: build tools/fetch_calibrate.hip and run it once per counter; prints counter bytes (raw, 1 KB = 1024 B units) per kernel.
set -e
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cal
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o gpurun_out/cal/fetch_calibrate
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/cal/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/cal/$c -- gpurun_out/cal/fetch_calibrate > gpurun_out/cal/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"gpurun_out/cal/{c}/*/*counter_collection.csv"))
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0]; tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot: out.setdefault(k, {})[c] = tot[k] / n[k] * 1024.0
B = 512 << 20
for k, d in sorted(out.items()):
    print(f"{k:16s} FETCH {d.get('FETCH_SIZE', 0)/1e6:9.1f} MB ({d.get('FETCH_SIZE', 0)/B:5.2f} x 512 MiB)   WRITE {d.get('WRITE_SIZE', 0)/1e6:9.1f} MB ({d.get('WRITE_SIZE', 0)/B:5.2f} x)")
json.dump({"buffer_bytes": B, "counters": out}, open("gpurun_out/cal/fetch_calibration.json", "w"), indent=1)
PY
