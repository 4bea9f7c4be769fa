# HBM traffic of the wd9 probe variants (FETCH_SIZE pass only; x2 correction as in the guide)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r04/pmc_probe -o run --output-format csv -- scripts/wd9_probe 2 3 > gpurun_out/r04/pmc_probe.log 2>&1
python - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04/pmc_probe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "wd9_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:110], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k[0][-70:], "grid", k[1], "launches", len(v), "fetch MB x2:", round(sum(v) / len(v) * 64 * 2 / 1e6, 1))
PY
