# (historical: needs the intermediate build with bench.py --store-nt; record: profiles/r04_tail_store_ab.txt)
# the opt-in wd9 tail (whole-line stores since it was written) with and without the non-temporal hint; and against the shipped two-wave tail
run() { timeout 300 python bench.py --steps 60 --warmup 5 "$@" --no-cpu-baseline --no-roofline --no-micro --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'])
"; }
for rep in 1 2; do run --wd9-mode 9 --store-nt 1; run --wd9-mode 13 --store-nt 0; run --wd9-mode 13 --store-nt 1; done
run --wd9-mode 9 --store-nt 1 --serial-detectors; run --wd9-mode 13 --store-nt 0 --serial-detectors; run --wd9-mode 13 --store-nt 1 --serial-detectors
