# the other workloads (bench.py --config 1 / 3 / 4) with and without the persistent kernels
mkdir -p gpurun_out/r04
run() { timeout 300 python bench.py --steps 40 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for c in 1 4 3; do
  for m in 0 1 5 0 5; do run --config $c --wd9-mode $m; done
done
