#!/bin/bash
# round 5: res4 bottleneck tails fused (default) vs conv2 (wd9 3x3) + conv3 (ring 1x1 with residual) as two launches
mkdir -p gpurun_out
{
echo "# bench.py --steps 60: pairs/s, ms per step"
for rep in 1 2 3; do
  for flags in "" "--no-tail-fusion"; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power $flags 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default $flags', d['value'], d['ms_per_step'])"
  done
done
for flags in "--serial-detectors" "--serial-detectors --no-tail-fusion"; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power $flags 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$flags', d['value'], d['ms_per_step'])"
done
} > gpurun_out/r05_pipeline_ab_tailfusion.txt 2>&1
cat gpurun_out/r05_pipeline_ab_tailfusion.txt
