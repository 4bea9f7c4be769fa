#!/bin/bash
# round 6: whole GPU suite + smoke + the bench line with the driver's flags (value_windows) twice + a longer run
O=gpurun_out/r06_check
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 > $O/bench_driver_$i.json; python -c "
import json; d = json.load(open('$O/bench_driver_$i.json')); print('driver-flags', d['value'], d['ms_per_step'], d['value_windows'], d.get('sclk_mhz_windows'))"; done
timeout 600 python bench.py --steps 250 --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 > $O/bench_250.json; python -c "
import json; d = json.load(open('$O/bench_250.json')); print('250 steps', d['value'], d['ms_per_step'], d['value_windows'], d.get('sclk_mhz_windows'))"
