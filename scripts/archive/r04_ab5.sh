# does the power sampler of bench.py perturb the timed region?  same box, alternating
run() { timeout 300 python bench.py --steps 100 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'], d.get('power'))
"; }
for i in 1 2; do run --no-power; run; done
