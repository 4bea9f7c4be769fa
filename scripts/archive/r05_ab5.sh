#!/bin/bash
# round 5: pipeline schedule knobs next to the ring kernel (stagger stage of the second detector)
mkdir -p gpurun_out
{
echo "# bench.py --steps 60: pairs/s, ms per step"
for rep in 1 2; do
  for flags in "--stagger 3" "--stagger 2" "--stagger 4" "--stagger 5" "--stagger 0"; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power $flags 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$flags', d['value'], d['ms_per_step'])"
  done
done
for flags in "--config 4 --stagger 3" "--config 4 --stagger 2" "--config 4 --stagger 4" "--config 4 --stagger 5"; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power $flags 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$flags', d['value'], d['ms_per_step'])"
done
} > gpurun_out/r05_pipeline_ab_stagger.txt 2>&1
cat gpurun_out/r05_pipeline_ab_stagger.txt
