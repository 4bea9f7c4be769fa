# (historical: needs the intermediate build with bench.py --store-nt; record: profiles/r04_tail_store_ab.txt)
# non-temporal output stores in the fused bottleneck tail (does streaming the 210 MB output past the L2 stop the shortcut re-fetch?)
mkdir -p gpurun_out/r04_nt
O=gpurun_out/r04_nt
python -m pytest tests/test_ops_gpu.py -q -k "bottleneck_tail" 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for s in 0 2 3 0 2 3; do run --store-nt $s; done | tee $O/ab.txt
for s in 0 2 3; do run --store-nt $s --serial-detectors; done | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 0 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_$s -o run --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors --store-nt $s > $O/stats_$s.log 2>&1
  grep -h "conv3x3_wd_kernel<1, 4, 4, 4, 0, 2>\|conv_igemm2_kernel<128, 128" $O/stats_$s/run_kernel_stats.csv | cut -c1-120 | tee -a $O/ab.txt
  for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$s -o run --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors --store-nt $s > $O/pmc_$s.log 2>&1
  python - <<PY | tee -a $O/ab.txt
import csv, glob
for f in glob.glob("$O/pmc_${c}_$s/**/run_counter_collection.csv", recursive=True):
    for kn in ("conv3x3_wd_kernel<1, 4, 4, 4, 0, 2>", "conv_igemm2_kernel<128, 128"):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f)):
            if kn in r["Kernel_Name"] and r["Counter_Name"] == "$c":
                tot += float(r["Counter_Value"]); n += 1
        print("store-nt $s", kn, "$c KiB per launch (uncorrected):", round(tot / max(n, 1)), "launches", n)
PY
  done
done
