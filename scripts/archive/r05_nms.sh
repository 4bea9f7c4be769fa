#!/bin/bash
# round 5: batched NMS - in-order input skips the sorting network: tests, pipeline A/B, kernel times under rocprof
mkdir -p gpurun_out/r05_nms
O=gpurun_out/r05_nms
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "nms or rpn or boxhead or proposal" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for s in 1 1; do run --nms-presorted $s; done > $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors"
for s in 1; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$s -o run --output-format csv -- $B --nms-presorted $s > $O/stats_$s.log 2>&1
grep -h "nms_" $O/stats_$s/*/run_kernel_stats.csv $O/stats_$s/run_kernel_stats.csv 2>/dev/null | cut -c1-150 > $O/kernels_$s.txt
python - <<PY >> $O/kernels_$s.txt
import csv, glob
for f in glob.glob("$O/stats_$s/**/run_kernel_trace.csv", recursive=True):
    d = {}
    for r in csv.DictReader(open(f)):
        if "nms_" in r["Kernel_Name"]:
            d.setdefault(r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("(") else 0][:40] + " grid " + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(d.items()):
        big = [x for x in v if x > sum(v) / len(v)] or v
        print(k, "launches", len(v), "us: min %.1f max %.1f" % (min(v), max(v)))
PY
done
cat $O/kernels_0.txt $O/kernels_1.txt
