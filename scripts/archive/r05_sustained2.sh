#!/bin/bash
# round 5, final kernels: bench.py 250 / 1000 / 250 steps back to back on one box (drift under load?)
mkdir -p gpurun_out
{
echo "# final round-5 kernels: python bench.py --steps N --warmup 5 --no-cpu-baseline --no-roofline --no-micro, back to back on one box"
for n in 250 1000 250; do
  timeout 300 python bench.py --steps $n --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d.get('power', {})
print('steps $n:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', p.get('board_w_median'), 'W median', p.get('sclk_mhz_median'), 'MHz')"
done
} > gpurun_out/r05_sustained_final.txt 2>&1
cat gpurun_out/r05_sustained_final.txt
