#!/bin/bash
# round 5: pipeline A/B of the conv dispatch policies (bench.py, 60 steps each, interleaved).  usage: bash scripts/r05_ab3.sh TAG "9 41 73" [extra bench flags]
TAG=${1:-x}; POLS=${2:-"9 41 73"}; shift; shift
mkdir -p gpurun_out
{
echo "# bench.py --steps 60 $@: pairs/s, ms per step"
for rep in 1 2 3; do
  for pol in $POLS; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --conv-policy $pol "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--conv-policy $pol', d['value'], d['ms_per_step'])"
  done
done
} > gpurun_out/r05_pipeline_ab_$TAG.txt 2>&1
cat gpurun_out/r05_pipeline_ab_$TAG.txt
