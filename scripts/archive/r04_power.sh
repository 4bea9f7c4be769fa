# Is the frame-pair pipeline power-limited?  Sample rocm-smi (power, sclk, mclk, temperature, use) every 0.25 s while bench.py runs.
mkdir -p gpurun_out/r04_power
O=gpurun_out/r04_power
which rocm-smi amd-smi > $O/tools.txt 2>&1
rocm-smi --showpower --showclocks --showuse --showtemp --showmaxpower > $O/idle.txt 2>&1
sample() {  # $1 = tag
  ( while true; do date +%s.%N; rocm-smi --showpower --showclocks --showuse --json 2>/dev/null; sleep 0.2; done ) > $O/samples_$1.txt 2>&1 &
  SP=$!
  timeout 300 python bench.py --steps 250 --warmup 10 --no-cpu-baseline --no-roofline --no-micro ${@:2} > $O/bench_$1.log 2>&1
  kill $SP
  grep '^{' $O/bench_$1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"
}
sample two_streams
sample serial --serial-detectors
python - <<'PY'
import json, re, glob
for f in sorted(glob.glob("gpurun_out/r04_power/samples_*.txt")):
    txt = open(f).read()
    recs = []
    for m in re.finditer(r"^\{.*\}$", txt, re.M):
        try: recs.append(json.loads(m.group(0)))
        except Exception: pass
    print(f, "samples", len(recs))
    if recs:
        keys = set()
        for r in recs:
            for card, v in r.items():
                if isinstance(v, dict): keys |= set(v.keys())
        print(" keys:", sorted(keys))
        for k in sorted(keys):
            vals = []
            for r in recs:
                for card, v in r.items():
                    if isinstance(v, dict) and k in v:
                        mm = re.search(r"[-+]?\d+(\.\d+)?", str(v[k]))
                        if mm: vals.append(float(mm.group(0)))
            if vals:
                vals.sort()
                print("  %-50s n=%d min %.1f median %.1f p90 %.1f max %.1f" % (k[:50], len(vals), vals[0], vals[len(vals)//2], vals[int(len(vals)*0.9)], vals[-1]))
PY
