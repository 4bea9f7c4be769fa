#!/bin/bash
# round 5: the opt-in wd9 fused tail once more, next to the ring 1x1 kernel (decides whether csrc/conv_wd9_tail.h stays in the library)
mkdir -p gpurun_out
{
echo "# bench.py --steps 60: pairs/s, ms per step"
for rep in 1 2; do
  for flags in "" "--wd9-mode 13" "--wd9-mode 13 --wd9-tail-wgs 128" "--wd9-mode 13 --wd9-tail-wgs 192"; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power $flags 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default $flags', d['value'], d['ms_per_step'])"
  done
done
} > gpurun_out/r05_pipeline_ab_wd9tail.txt 2>&1
cat gpurun_out/r05_pipeline_ab_wd9tail.txt
