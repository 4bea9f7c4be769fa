for round in 1 2; do
for w in 256 384 512 768; do
PE_RING_WGS=$w python scripts/lab/bench_with_lib.py product --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('round $round ring wgs $w  %8.2f pairs/s  %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
