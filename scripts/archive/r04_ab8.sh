# (ran on an intermediate build that carried a store-policy switch in every kernel: csrc of the working tree before commit "fused tail: whole-line non-temporal stores"; record: profiles/r04_tail_store_ab.txt)
# which kernels should stream their outputs (non-temporal hint)?  bench.py --store-nt MASK (csrc/common.h g_store_policy), same box, alternating
mkdir -p gpurun_out/r04_nt2
O=gpurun_out/r04_nt2
python -m pytest tests/test_ops_gpu.py -q -k "bottleneck_tail or conv or roi_align or bneck" 2>&1 | tail -2 | tee $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for rep in 1 2; do for s in 0 1 5 9 17 33 61; do run --store-nt $s; done; done | tee $O/ab.txt
for s in 1 61; do run --store-nt $s --serial-detectors; done | tee -a $O/ab.txt
