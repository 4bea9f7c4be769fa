# RPN head on the one-wave structure (wd9 mode bit 3): parity tests, then A/B inside the pipeline (same box, alternating)
mkdir -p gpurun_out/r04_head
python -m pytest tests/test_ops_gpu.py -q -k "rpn_head or wd9" 2>&1 | tail -4 > gpurun_out/r04_head/tests.txt
cat gpurun_out/r04_head/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for m in 1 9 1 9; do run --wd9-mode $m; done | tee gpurun_out/r04_head/ab.txt
for m in 1 9; do run --wd9-mode $m --serial-detectors; done | tee -a gpurun_out/r04_head/ab.txt
for c in 1 4 3; do for m in 1 9; do run --config $c --wd9-mode $m; done; done | tee -a gpurun_out/r04_head/ab.txt
