#!/bin/bash
# round 5: ROIAlign wave-uniform form vs the per-lane form - identity test, kernel time (rocprof), pipeline A/B, one call
mkdir -p gpurun_out/r05_roi
O=gpurun_out/r05_roi
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "roi_align" 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for s in 0 1 0 1 0 1; do run --roi-fast $s; done > $O/ab.txt
for s in 0 1; do run --roi-fast $s --config 3; done >> $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors"
for s in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$s -o run --output-format csv -- $B --roi-fast $s > $O/stats_$s.log 2>&1
  grep -h "roi_" $O/stats_$s/*/run_kernel_stats.csv $O/stats_$s/run_kernel_stats.csv 2>/dev/null | cut -c1-160 > $O/roi_stats_$s.txt
done
cat $O/roi_stats_*.txt
{ echo "# ROIAlign (fp16, C = 256, 32 x 1000 proposals, 4 levels): per-lane form (0) vs wave-uniform form (1)"; echo "## tests"; cat $O/tests.txt; echo "## bench.py --steps 60 (configs[2]; last two: configs[3]): pairs/s, ms per step"; cat $O/ab.txt; echo "## rocprofv3 --kernel-trace --stats, serial detectors, 5 steps: --roi-fast 0"; cat $O/roi_stats_0.txt; echo "## --roi-fast 1"; cat $O/roi_stats_1.txt; } > gpurun_out/r05_roi_fast_ab.txt
