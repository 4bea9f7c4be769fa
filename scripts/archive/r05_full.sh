#!/bin/bash
# round 5: whole GPU suite + smoke + a short bench (checkpoint after the ROIAlign / ProbEn / NMS / preprocess kernels)
mkdir -p gpurun_out/r05_full
O=gpurun_out/r05_full
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors > $O/stats.log 2>&1
grep -h "preprocess\|nms_\|proben\|roi_\|rpn_\|stem" $O/stats/*/run_kernel_stats.csv $O/stats/run_kernel_stats.csv 2>/dev/null | cut -c1-150 > $O/kernels.txt
cat $O/kernels.txt
