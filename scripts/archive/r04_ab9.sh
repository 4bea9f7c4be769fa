# (historical: needs the intermediate build with bench.py --store-nt; record: profiles/r04_tail_store_ab.txt)
# the fused tail's output stores, same box: --store-nt 2 = r03 stores (16-byte pieces), 1 = whole lines + nt (shipped), 5 = + nt in conv_igemm2.hip's epilogues
mkdir -p gpurun_out/r04_nt3
O=gpurun_out/r04_nt3
python -m pytest tests/test_ops_gpu.py -q -k "bottleneck_tail" 2>&1 | tail -1 | tee $O/tests.txt
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for rep in 1 2 3; do for s in 2 1 5; do run --store-nt $s; done; done | tee $O/ab.txt
for s in 2 1 5; do run --store-nt $s --serial-detectors; done | tee -a $O/ab.txt
for c in 1 4 3; do for s in 2 5; do run --config $c --store-nt $s; done; done | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 2 5; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$s -o run --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors --store-nt $s > $O/stats_$s.log 2>&1
  grep -h "conv3x3_wd_kernel<1, 4, 4, 4, 0, 2>\|conv_igemm2_kernel<128, 128" $O/stats_$s/run_kernel_stats.csv | cut -c1-120 | sed "s/^/store-nt $s /" | tee -a $O/ab.txt
done
