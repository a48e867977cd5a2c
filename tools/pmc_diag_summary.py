"""Per-launch means of every counter collected by tools/pmc_diag.sh for one kernel, plus the kernel's
average duration from the stats pass.

usage: python tools/pmc_diag_summary.py <tag> <kernel substring>"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, kernel = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Name"]:
                    print(f"# stats: {row['Name'][:90]}  calls {row['Calls']}  avg {float(row['AverageNs']) / 1e3:.2f} us  "
                          f"min {float(row['MinNs']) / 1e3:.2f}  max {float(row['MaxNs']) / 1e3:.2f}")
    vals = {}
    meta = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                        if k in row:
                            meta[k] = row[k]
    print("# launch:", " ".join(f"{k}={v}" for k, v in sorted(meta.items())))
    for k in sorted(vals):
        v = vals[k]
        print(f"{k:42s} mean {sum(v) / len(v):14.5e}  min {min(v):12.5e}  max {max(v):12.5e}  launches {len(v)}")


if __name__ == "__main__":
    main()
