#!/bin/bash
# round 5, GPU call 1: the 1x1 ring kernel - bits, per-launch time, pipeline A/B
mkdir -p gpurun_out
{
timeout 300 python scripts/r05_ring.py
echo "# pipeline A/B (bench.py, 60 steps): --conv-policy 9 (r04 default) / 41 (+ ring for res4 conv1) / 73 (+ ring for every eligible 1x1)"
for rep in 1 2; do
  for pol in 9 41 73; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --conv-policy $pol 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--conv-policy $pol', d['value'], d['ms_per_step'])"
  done
done
for pol in 9 41; do
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors --conv-policy $pol 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--serial-detectors --conv-policy $pol', d['value'], d['ms_per_step'])"
done
} > gpurun_out/r05_ring_1.txt 2>&1
tail -50 gpurun_out/r05_ring_1.txt
