#!/bin/bash
# lab: bench.py on several builds of the library, alternating, on one box:  scripts/lab/pipeline_ab.sh 100 product exp_x [exp_y ...]
STEPS=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    python scripts/lab/bench_with_lib.py $v --steps $STEPS --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('round $round  %-14s %8.2f pairs/s  %.3f ms  windows %s  sclk %s' % ('$v', d['value'], d['ms_per_step'], d.get('value_windows'), d.get('sclk_mhz_windows')))"
  done
done
