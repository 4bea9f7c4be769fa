#!/bin/bash
# round 5, final kernels: stagger stage of the two detector streams (configs[2])
mkdir -p gpurun_out
{
echo "# bench.py --steps 60 (configs[2]), final round-5 kernels: pairs/s, ms per step"
for rep in 1 2; do
  for st in 2 3 4 5; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --stagger $st 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--stagger $st', d['value'], d['ms_per_step'])"
  done
done
} > gpurun_out/r05_pipeline_ab_stagger_final.txt 2>&1
cat gpurun_out/r05_pipeline_ab_stagger_final.txt
