#!/bin/bash
# round 6: the other BASELINE workloads + host feeding + a sustained run, one box (profiles/r06_bench_configs.jsonl, r06_sustained.txt)
O=gpurun_out/r06_configs
mkdir -p $O
F="--no-cpu-baseline --no-roofline --no-micro"
: > $O/bench_configs.jsonl
for c in 1 3 4; do timeout 900 python bench.py --config $c --steps 100 --warmup 5 $F 2>/dev/null | tail -1 >> $O/bench_configs.jsonl; done
timeout 900 python bench.py --config 2 --feed host --steps 100 --warmup 5 $F 2>/dev/null | tail -1 >> $O/bench_configs.jsonl
python - <<'E'
import json
for l in open("gpurun_out/r06_configs/bench_configs.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:70], d["value"], d["unit"], d["ms_per_step"], d["config"].get("input_residency", "")[:30])
E
: > $O/sustained.txt
for s in 250 1000 250; do timeout 900 python bench.py --steps $s --warmup 5 $F 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('steps', d['steps'], 'pairs/s', d['value'], 'ms', d['ms_per_step'], 'windows', d['value_windows'], 'W', d['power'].get('board_w_median'), 'sclk', d.get('sclk_mhz_windows'))" | tee -a $O/sustained.txt; done
