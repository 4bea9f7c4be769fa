# ROIAlign processing order ("(level, y, x) ROI ordering / XCD-affine mapping"): parity, A/B, kernel time, FETCH_SIZE
mkdir -p gpurun_out/r04_roi
O=gpurun_out/r04_roi
python -m pytest tests/test_ops_gpu.py -q -k "roi_align" 2>&1 | tail -3 > $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for s in 0 1 0 1; do run --roi-sort $s; done > $O/ab.txt
for s in 0 1; do run --roi-sort $s --serial-detectors; done >> $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --serial-detectors"
for s in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$s -o run --output-format csv -- $B --roi-sort $s > $O/stats_$s.log 2>&1
  grep -h "roi_" $O/stats_$s/*/run_kernel_stats.csv $O/stats_$s/run_kernel_stats.csv 2>/dev/null | cut -c1-160 > $O/roi_stats_$s.txt
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$s -o run --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-micro --serial-detectors --roi-sort $s > $O/pmc_$s.log 2>&1
  python - <<PY > $O/roi_fetch_$s.txt
import csv, glob
for f in glob.glob("$O/pmc_$s/**/run_counter_collection.csv", recursive=True):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "roi_align_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
    print("roi_align FETCH_SIZE KiB per launch (uncorrected):", tot / max(n, 1), "launches", n)
PY
done
cat $O/tests.txt $O/ab.txt $O/roi_stats_*.txt $O/roi_fetch_*.txt
