#!/bin/bash
# round 5, final record: profile collection, same-box A/B of the late-round kernels, the other configs, the whole GPU suite
mkdir -p gpurun_out/r05_final
O=gpurun_out/r05_final
bash scripts/collect_profiles.sh r05
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
{ for rep in 1 2; do run --roi-fast 0 --nms-presorted 0; run; done; } > $O/ab.txt
cat $O/ab.txt
for c in 1 3 4; do timeout 300 python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1; done > $O/configs.jsonl
timeout 300 python bench.py --feed host --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 >> $O/configs.jsonl
cut -c1-200 $O/configs.jsonl
timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_driver_flags.json
cut -c1-160 $O/bench_driver_flags.json
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gpu_tests.txt
cat $O/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 >> $O/gpu_tests.txt
tail -1 $O/gpu_tests.txt
