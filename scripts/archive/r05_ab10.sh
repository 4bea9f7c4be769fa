#!/bin/bash
# round 5: stagger stage for the two-R50 configuration (configs[4]) with the final kernels
mkdir -p gpurun_out
{
echo "# bench.py --config 4 --steps 60: pairs/s, ms per step"
for rep in 1 2 3; do
  for st in 3 2 1; do
    timeout 300 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --stagger $st 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--stagger $st', d['value'], d['ms_per_step'])"
  done
done
} > gpurun_out/r05_pipeline_ab_stagger_c4.txt 2>&1
cat gpurun_out/r05_pipeline_ab_stagger_c4.txt
