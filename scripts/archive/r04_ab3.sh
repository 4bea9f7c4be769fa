# lockstep (stagger 0) vs staggered two-stream pipeline with half-chip persistent kernels
mkdir -p gpurun_out/r04
run() { timeout 200 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
run --wd9-mode 1
run --wd9-mode 1 --stagger 0
run --wd9-mode 5 --wd9-tail-wgs 128 --stagger 0
run --wd9-mode 5 --wd9-tail-wgs 128 --wd9-wgs 128 --stagger 0
run --wd9-mode 5 --wd9-tail-wgs 128
run --wd9-mode 5 --wd9-tail-wgs 136 --wd9-wgs 128 --stagger 0
run --wd9-mode 0 --stagger 0
run --wd9-mode 1
