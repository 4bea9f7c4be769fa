"""Summarise rocprofv3 --pmc output: python scripts/pmc_summary.py <dir> [name-filter]
Per kernel name (template arguments kept, argument list cut): launches, mean duration, mean of every counter, and
the derived figures used in DESIGN.md (effective clock = GRBM_GUI_ACTIVE / duration; MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  [both in shader cycles])."""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name[5:] if name.startswith("void ") else name
    depth, out = 0, []
    for ch in name:           # cut the argument list: the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[-70:]


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"]
            if flt and flt not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    for k in sorted(acc):
        d = sorted(dur[k].values())
        dm = d[len(d) // 2]
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        print(f"{k}  launches={len(d)} median_dur_ms={dm * 1e3:.4f}")
        for n in sorted(c):
            print(f"    {n:28s} {c[n]:.4g}")
        if "GRBM_GUI_ACTIVE" in c:
            print(f"    -> effective clock {c['GRBM_GUI_ACTIVE'] / 8 / dm / 1e9:.3f} GHz (GRBM_GUI_ACTIVE is summed over the 8 XCDs)")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            print(f"    -> MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):.3f} of CU-busy cycles")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            print(f"    -> MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * c['GRBM_GUI_ACTIVE']):.3f} of (GUI_ACTIVE per XCD x 1024 SIMDs)")
        if "SQ_WAVE_CYCLES" in c:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS"):
                if n in c:
                    print(f"    -> {n} / SQ_WAVE_CYCLES = {c[n] / c['SQ_WAVE_CYCLES']:.3f}")


if __name__ == "__main__":
    main()
