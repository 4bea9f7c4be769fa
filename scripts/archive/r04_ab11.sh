# does the best stagger stage move now that the res4 tail is 20 % shorter?  (--stagger S: detector 1 starts when detector 0 has finished res S; 0 = plain two streams)
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for rep in 1 2; do for s in 3 2 4 5 0; do run --stagger $s; done; done
