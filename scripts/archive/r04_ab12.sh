# the 1x1 class is the dominant kernel now: re-check its dispatch policy inside the pipeline (bench.py --conv-policy BITS; default 9;
# +2 = 256-row tiles for large 1x1 launches, +4 = two-stage pipeline in the generic 1x1 kernel)
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for rep in 1 2; do for p in 9 11 13 15; do run --conv-policy $p; done; done
for p in 9 11 13 15; do run --conv-policy $p --serial-detectors; done
