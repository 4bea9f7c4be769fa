#!/bin/bash
# PMC view of scripts/phase_overlap_probe's six dispatches (see its PMC mode): effective clock and MFMA-busy share per dispatch -
# does the clock drop when the matrix set and the memory set share the CUs (power), or does the matrix pipe just sit idle (issue)?
set -u
OUT=${1:-gpurun_out/r03/overlap_pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/op; timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY -d /tmp/op -o run --output-format csv -- $R/scripts/phase_overlap_probe 8192 5200 > $R/$OUT/run.log 2>&1
f=$(find /tmp/op -name "*counter_collection.csv" | head -1)
python - "$f" > $R/$OUT/overlap_pmc.txt <<'PY'
import csv, sys, collections
csv.field_size_limit(1 << 30)
d = collections.defaultdict(dict); t = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "probe" not in r["Kernel_Name"]: continue
    k = int(r["Dispatch_Id"]); d[k][r["Counter_Name"]] = d[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    t[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
names = ["matrix set alone (full diet)", "memory set alone", "both at once (full diet)", "matrix set alone (no weight loads)", "both at once (no weight loads)", "phase-locked (all 8 waves: matrix half, then memory half)"]
print("dispatch                                                   ms    clock GHz  MFMA busy  wait_inst  wait_any  vmem-active  (shares of SQ_WAVE_CYCLES)")
for i, k in enumerate(sorted(d)):
    c = d[k]; dur = t[k]; wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{names[i] if i < len(names) else k:58s} {dur*1e3:6.3f}  {c.get('GRBM_GUI_ACTIVE',0)/8/dur/1e9:8.2f}  {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(128*c.get('GRBM_GUI_ACTIVE',1),1):9.3f}  "
          f"{c.get('SQ_WAIT_INST_ANY',0)/wc:9.2f} {c.get('SQ_WAIT_ANY',0)/wc:9.2f} {c.get('SQ_ACTIVE_INST_VMEM',0)/wc:11.3f}")
PY
cat $R/$OUT/overlap_pmc.txt
