# A/B of the kernel generations inside the whole pipeline (same box, same run): bench.py --wd9-mode x two-stream / serial
mkdir -p gpurun_out/r04
for mode in ${MODES:-0 1 5}; do
  for ser in "" "--serial-detectors"; do
    timeout 200 python bench.py --steps ${STEPS:-60} --warmup 5 --wd9-mode $mode $ser --no-cpu-baseline --no-roofline --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wd9-mode $mode $ser', d['value'], d['ms_per_step'])
"
  done
done
