#!/bin/bash
# round 5: ring kernel - bits / per-launch time / ablations / pipeline A/B.  usage: bash scripts/r05_ab2.sh TAG
TAG=${1:-x}
mkdir -p gpurun_out
{
(cd scripts && timeout 300 python r05_ring.py && timeout 300 python r05_ring_abl.py | grep -v "wgs 512")
echo "# pipeline A/B (bench.py, 60 steps): --conv-policy 9 (r04 default) / 41 (+ ring for res4 conv1) / 73 (+ ring for every eligible 1x1)"
for rep in 1 2; do
  for pol in 9 41 73; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-micro --no-power --conv-policy $pol 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--conv-policy $pol', d['value'], d['ms_per_step'])"
  done
done
} > gpurun_out/r05_ring_$TAG.txt 2>&1
tail -60 gpurun_out/r05_ring_$TAG.txt
