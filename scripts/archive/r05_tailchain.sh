#!/bin/bash
# round 5: the serial ends of the step (ProbEn fusion, NMS chain): tests, kernel times under rocprof, pipeline
mkdir -p gpurun_out/r05_tailchain
O=gpurun_out/r05_tailchain
timeout 900 python -m pytest tests/test_proben_gpu.py tests/test_ops_gpu.py -q -x -k "proben or roi_align or nms or lds_staged or hip_matches or binary or full_size" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for s in 1 2; do run; done > $O/ab.txt
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $B > $O/stats.log 2>&1
grep -h "proben\|nms_\|rpn_s\|roi_" $O/stats/*/run_kernel_stats.csv $O/stats/run_kernel_stats.csv 2>/dev/null | cut -c1-150 > $O/kernels.txt
cat $O/kernels.txt
