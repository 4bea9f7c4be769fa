#!/bin/bash
# is the guide's "FETCH_SIZE x2" correction right for the tail kernel's shortcut reads (a lane owns 64 contiguous
# bytes, requested as 16-byte pieces)?  Runs scripts/stride_probe - every kernel of it reads a KNOWN byte count (1680 MiB per
# launch for sweep<> and sweep_lane64<>) - under one FETCH_SIZE pass and prints counter KiB / known KiB per kernel.
set -u
OUT=${1:-gpurun_out/r03/fetch_cal}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o run --output-format csv -- $R/scripts/stride_probe > $R/$OUT/stride_probe.log 2>&1
f=$(find /tmp/fc -name "*counter_collection.csv" | head -1)
python - "$f" > $R/$OUT/fetch_calibration.txt <<'PY'
import csv, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
known = 1680 * 1024.0   # KiB read per launch by sweep<> and sweep_lane64<> (scripts/stride_probe.hip)
print("FETCH_SIZE (KiB, as reported) / bytes the kernel is known to read; 0.5 = the guide's 'reports half' case")
for k, v in acc.items():
    if "sweep_rw" in k:
        continue
    m = sum(v) / len(v)
    print(f"{k[:60]:60s} launches {len(v):3d}  FETCH_SIZE {m / 1024:9.1f} MiB  ratio to 1680 MiB: {m / known:.3f}  (min {min(v) / known:.3f} max {max(v) / known:.3f})")
PY
cat $R/$OUT/fetch_calibration.txt
